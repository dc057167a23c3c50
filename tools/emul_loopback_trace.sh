# kernel trace of a rank's frame at M emulated ranks with the loopback collective (every rank's slot = this rank's slot):
# the unpack then has the volume of a real M-rank run.  Usage: bash tools/emul_loopback_trace.sh M
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=${1:-8}
OUT=$R/gpurun_out/emul
mkdir -p $OUT
TR_EMULATE_LOOPBACK=1 rocprofv3 --kernel-trace --stats -d $OUT/loop$M -o t -- python3 $R/bench.py --emulate-ranks $M --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/loop$M.log 2>&1
python3 $R/tools/dbstats.py $OUT/loop$M > $OUT/loop$M.txt
cat $OUT/loop$M.txt
