cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/emul
mkdir -p $OUT
for m in 2 4 8; do
  TR_NO_GATHER=1 python3 $R/bench.py --emulate-ranks $m --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('M', $m, 'shard frame, no exchange: ms', d['ms_per_step'])"
  python3 $R/bench.py --emulate-ranks $m --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('M', $m, 'shard frame + exchange (1-rank RCCL group, one slot unpacked): ms', d['ms_per_step'])"
  TR_EMULATE_LOOPBACK=1 python3 $R/bench.py --emulate-ranks $m --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('M', $m, 'shard frame + exchange (loopback: M slots unpacked): ms', d['ms_per_step'])"
done > $OUT/emulated.txt
rocprofv3 --kernel-trace --stats -d $OUT/trace8 -o t -- python3 $R/bench.py --emulate-ranks 8 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/trace8.log 2>&1
cat $OUT/emulated.txt
