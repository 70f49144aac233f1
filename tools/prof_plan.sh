export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rm -rf /tmp/pp_stats
N_PREC=300000 REPS=1 CHUNKS=262144 rocprofv3 --kernel-trace --stats -d /tmp/pp_stats -o p -- python $REPO/tools/bench_h2h.py > /tmp/pp_stats.log 2>&1
tail -2 /tmp/pp_stats.log
python $REPO/tools/rocpd_summary.py /tmp/pp_stats/p_results.db | grep -v "^#" | awk -F, '{n=$1; sub(/\(.*/,"",n); printf "%-60s calls %5s avg_us %9.1f\n", substr(n,1,60), $(NF-5), $(NF-3)/1e3}' | head -24
