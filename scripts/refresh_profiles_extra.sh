#!/bin/bash
# Run on the GPU box from the repo root: kernel traces of the canonical view, the fitting iteration (world / canonical space) and
# the full bench line, summaries only (the rocprofv3 databases are deleted) under gpurun_out/r01b (copy into profiles/ what should be judged).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r01b; mkdir -p $O
rocprofv3 --kernel-trace --stats -d /tmp/p_can -o can -- python scripts/canonical_bench.py > $O/can.log 2>&1
python scripts/rocpd_summary.py $(ls /tmp/p_can/*results.db /tmp/p_can/*/*results.db 2>/dev/null | head -1) $O/canonical_view_trace.md > /dev/null
rocprofv3 --kernel-trace --stats -d /tmp/p_fitc -o fitc -- python scripts/train_bench.py 10 canonical > $O/fitc.log 2>&1
python scripts/rocpd_summary.py $(ls /tmp/p_fitc/*results.db /tmp/p_fitc/*/*results.db 2>/dev/null | head -1) $O/fit_canonical_trace.md > /dev/null
rocprofv3 --kernel-trace --stats -d /tmp/p_fit -o fit -- python scripts/train_bench.py 10 > $O/fit.log 2>&1
python scripts/rocpd_summary.py $(ls /tmp/p_fit/*results.db /tmp/p_fit/*/*results.db 2>/dev/null | head -1) $O/fit_trace.md > /dev/null
python bench.py > $O/bench_full.log 2>&1; grep '^{"metric"' $O/bench_full.log | tail -1 > $O/bench.json
python scripts/train_bench.py 20 2>/dev/null | tail -1 > $O/fit_wall.txt; python scripts/train_bench.py 20 canonical 2>/dev/null | tail -1 > $O/fitc_wall.txt
python scripts/canonical_bench.py 2>/dev/null | tail -1 > $O/can_wall.txt
ls -la $O
