#!/bin/bash
cd /root/repo
bash tools/ab_env_long.sh PARQ_FUSE_SEAMS 1 0 2>&1 | tee gpurun_out/r06i_ab_seam_second_box.txt
