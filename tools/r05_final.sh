#!/bin/bash
# round 5: smoke() + the whole GPU suite (what the driver runs at round end)
cd /root/repo
out=/root/repo/gpurun_out/${1:-r05_final}
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 3000 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $out/pytest.log
