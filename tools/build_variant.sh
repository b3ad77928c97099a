#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -- an A/B build of the HIP library under
# invpref_kdd_2022_amd/variants/NAME.so (git-ignored, shipped by gpurun); select it with INVPREF_LIB=<path>.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
rm -rf /tmp/variant_$name; mkdir -p invpref_kdd_2022_amd/variants /tmp/variant_$name
for f in invpref_kernels invpref_step invpref_eval; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-pass-failed "$@" \
    -c invpref_kdd_2022_amd/csrc/$f.hip -o /tmp/variant_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/variant_$name/*.o -o invpref_kdd_2022_amd/variants/$name.so
echo built invpref_kdd_2022_amd/variants/$name.so
