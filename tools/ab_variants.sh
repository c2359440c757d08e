#!/bin/bash
# GPU-box helper: time the training-step kernels with each prebuilt library variant parq_amd/_C/variants/lib_v*.so
cd /root/repo
cp parq_amd/_C/libparq_hip.so /tmp/lib_base.so
for lib in /tmp/lib_base.so parq_amd/_C/variants/lib_v*.so; do
  cp $lib parq_amd/_C/libparq_hip.so
  echo "== $lib"
  tools/prof_train.sh 2>&1 | sed -n 2,2p | cut -c1-40,90-140
done
cp /tmp/lib_base.so parq_amd/_C/libparq_hip.so
