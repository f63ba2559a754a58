# rocprofv3 kernel trace of 5-key-frame calls (second lap): the kernels of the last two calls with start offsets and gaps.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04s}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o r -- python $R/scripts/experiments/small_calls.py ${2:-5} 2 ${4:-} 2>&1 | grep "K=" ) > $O/calls.log 2>&1
cat $O/calls.log
python scripts/experiments/trace_gaps.py $(find $O/prof -name "*kernel_trace.csv" | head -1) ${3:-50} > $O/gaps.txt
cat $O/gaps.txt
rm -rf $O/prof
