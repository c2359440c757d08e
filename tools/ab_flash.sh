#!/bin/bash
# GPU box: attention parity tests, then the default bench three times (cross-attention launch time from hipEvents)
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_decoder.py -m gpu -q -k "attention_split or golden or determinism or ragged" 2>&1 | tail -2
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-b32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('it/s %.1f  ms %.4f  flash %.1f us  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']))"; done
