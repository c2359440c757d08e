#!/bin/bash
# GPU box: does the code alignment of the cross-attention's stage loop matter?  (tools/bench_src/mx_energy.hip: a bare MFMA loop runs at
# 32.8 or at 49 cycles per MFMA depending on where its head falls in a 32-byte window.)  Rebuilds flash_split8.o with N s_nop in front
# of the loop, relinks the product objects into a scratch library and times the headline bench with it.
cd /root/repo
O=parq_amd/_C
cp $O/libparq_hip.so /tmp/libparq_hip.orig.so
for sh in ${SHIFTS:-0 1 2 3 4 5 6 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPARQ_F8_SHIFT=$sh -c parq_amd/csrc/flash_split8.hip -o /tmp/f8_$sh.o 2>/dev/null
  objs=$(ls $O/*.o | grep -v flash_split8.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libparq_hip.so $objs /tmp/f8_$sh.o 2>/dev/null
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-b32 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shift $sh: flash %.1f us  forward %.4f ms  %.0f it/s' % (d['roofline']['avg_launch_ms']*1000, d['ms_per_step'], d['value']))"
done
cp /tmp/libparq_hip.orig.so $O/libparq_hip.so
