cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/pf && mkdir -p $O/pf
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pf -- python3 $R/tools/_apply_compress_one.py > $O/pf.log 2>&1
tail -1 $O/pf.log
f=$(find $O/pf -name '*kernel_stats.csv' | head -1)
python3 $R/tools/_kstats.py $f 22
rm -rf $O/pf
