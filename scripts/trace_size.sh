# kernel trace of one warm fit + predict: bash scripts/trace_size.sh N n tag [option=value ...]   (TRACE_LIST=k: raw listing)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=$1; n=$2; tag=$3; shift 3
rm -rf gpurun_out/prof_tmp && mkdir -p gpurun_out/prof_tmp
rocprofv3 --kernel-trace -d gpurun_out/prof_tmp/kt -- python3 scripts/trace_one.py $N $n "$@" > gpurun_out/trace_${tag}.out 2> gpurun_out/trace_${tag}.err || exit 1
python3 scripts/trace_timeline.py $(find gpurun_out/prof_tmp/kt -name "*.db" | head -1) $TRACE_WINDOW_MS > gpurun_out/trace_${tag}_timeline.txt
rm -rf gpurun_out/prof_tmp
cat gpurun_out/trace_${tag}.out; head -40 gpurun_out/trace_${tag}_timeline.txt
