set -e
cd surf_amd/csrc
for f in ray_setup composite blend; do cp ../_obj/$f.o /tmp/$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSURF_SDF_OCC=1 -c sdf_mlp.hip -o /tmp/sdf_mlp.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs Spill|ScratchSize" 
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsurf_occ1.so /tmp/ray_setup.o /tmp/composite.o /tmp/blend.o /tmp/sdf_mlp.o
cd ../..
SURF_HIP_LIB=/tmp/libsurf_occ1.so python scripts/dbg_sdf.py 2>&1 | grep -E "bad idx|repeat" | cut -c1-300
