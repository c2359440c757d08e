#!/bin/bash
# Build the CURRENT source tree into parq_amd/_C/variants/lib_$1.so (extra hipcc flags in $2) without touching the main library.
set -e
cd /root/repo
mkdir -p parq_amd/_C/variants/obj_$1
for f in parq_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -c $f -o parq_amd/_C/variants/obj_$1/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o parq_amd/_C/variants/lib_$1.so parq_amd/_C/variants/obj_$1/*.o
rm -rf parq_amd/_C/variants/obj_$1
ls -la parq_amd/_C/variants/lib_$1.so
