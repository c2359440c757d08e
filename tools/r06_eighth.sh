#!/bin/bash
cd /root/repo
timeout 600 python tools/r06_inflight_profile.py 2>&1 | grep -v amdgpu.ids | head -16
