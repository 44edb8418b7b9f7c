cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python scripts/unet_train_bench.py 3 2 2>&1 | tail -2
O=gpurun_out/train; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/tr -- python scripts/unet_train_bench.py 2 2 > $O/tr.log 2>&1
python scripts/rocpd_summary.py $(ls $O/tr/*/*results.db | head -1) $O/trace_train.md | head -24
rm -rf $O/tr
