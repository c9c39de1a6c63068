"""HOCON-subset parser exposing the pyhocon getters the reference uses (runner.py:35-52, surf.py:19-21,
implicit_surface.py:54-62): nested ``{}`` blocks, ``key = value`` / ``key: value``, lists, ``#`` and ``//``
comments, numbers, booleans, quoted strings and bare strings (``<your output save path>``)."""
import re


class ConfigTree(dict):
    def _get(self, key, default=None):
        cur = self
        for part in str(key).split("."):
            if not isinstance(cur, dict) or not dict.__contains__(cur, part):
                return default
            cur = dict.__getitem__(cur, part)
        return cur

    def __getitem__(self, key):
        v = self._get(key, _MISSING)
        if v is _MISSING:
            raise KeyError(key)
        return v

    def __contains__(self, key):
        return self._get(key, _MISSING) is not _MISSING

    def get(self, key, default=None):
        return self._get(key, default)

    def get_int(self, key, default=None):
        v = self._get(key, default)
        return None if v is None else int(v)

    def get_float(self, key, default=None):
        v = self._get(key, default)
        return None if v is None else float(v)

    def get_bool(self, key, default=None):
        v = self._get(key, default)
        if isinstance(v, str):
            return v.lower() in ("true", "yes", "on")
        return None if v is None else bool(v)

    def get_list(self, key, default=None):
        v = self._get(key, default)
        return None if v is None else list(v)

    def get_string(self, key, default=None):
        v = self._get(key, default)
        return None if v is None else str(v)

    def get_config(self, key, default=None):
        return self._get(key, default)


_MISSING = object()
_TOKEN = re.compile(r"""\s*(?:(?P<com>(?:\#|//)[^\n]*)|(?P<punct>[{}\[\],=:])|"(?P<q>(?:[^"\\]|\\.)*)"|(?P<nl>\n)|(?P<bare>[^\s{}\[\],=:#"][^\n{}\[\],=:#]*))""")


def _scalar(tok):
    t = tok.strip()
    low = t.lower()
    if low in ("true", "yes", "on"):
        return True
    if low in ("false", "no", "off"):
        return False
    if low == "null":
        return None
    try:
        return int(t)
    except ValueError:
        pass
    try:
        return float(t)
    except ValueError:
        return t


def _tokens(text):
    pos, out = 0, []
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            raise ValueError(f"conf: cannot tokenize at {text[pos:pos + 30]!r}")
        pos = m.end()
        if m.group("com") is not None or m.group("nl") is not None:
            continue
        if m.group("punct"):
            out.append(("p", m.group("punct")))
        elif m.group("q") is not None:
            out.append(("v", m.group("q")))
        else:
            out.append(("b", m.group("bare")))
    return out


class _Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def take(self):
        tok = self.peek()
        self.i += 1
        return tok

    def obj(self, top=False):
        tree = ConfigTree()
        while True:
            kind, val = self.peek()
            if kind is None:
                if top:
                    return tree
                raise ValueError("conf: unexpected end inside {}")
            if (kind, val) == ("p", "}"):
                self.take()
                return tree
            if (kind, val) == ("p", ","):
                self.take()
                continue
            key = str(self.take()[1]).strip()
            kind, val = self.peek()
            if (kind, val) == ("p", "{"):
                self.take()
                v = self.obj()
            else:
                if (kind, val) in (("p", "="), ("p", ":")):
                    self.take()
                v = self.value()
            node = tree
            parts = key.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, ConfigTree())
            if isinstance(v, dict) and isinstance(node.get(parts[-1]), dict):
                dict.__getitem__(node, parts[-1]).update(v)
            else:
                dict.__setitem__(node, parts[-1], v)

    def value(self):
        kind, val = self.take()
        if (kind, val) == ("p", "{"):
            return self.obj()
        if (kind, val) == ("p", "["):
            out = []
            while True:
                k, v = self.peek()
                if (k, v) == ("p", "]"):
                    self.take()
                    return out
                if (k, v) == ("p", ","):
                    self.take()
                    continue
                out.append(self.value())
        if kind == "v":
            return val
        if kind == "b":
            return _scalar(val)
        raise ValueError(f"conf: unexpected token {val!r}")


def parse_string(text):
    toks = _tokens(text)
    if toks and toks[0] == ("p", "{"):
        p = _Parser(toks[1:])
        return p.obj()
    return _Parser(toks).obj(top=True)


def parse_file(path):
    with open(path) as f:
        return parse_string(f.read())


def from_dict(d):
    t = ConfigTree()
    for k, v in d.items():
        dict.__setitem__(t, k, from_dict(v) if isinstance(v, dict) else v)
    return t
