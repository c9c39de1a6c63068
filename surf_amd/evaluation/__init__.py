"""Offline evaluation: DTU Chamfer distance and mask / visibility based mesh cleaning."""
