#!/bin/bash
# Round-3 second batch: where the cull kernel's wave-cycles go (stamps), lookup emulation in the stream benchmark.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b
mkdir -p $OUT
cd $R
timeout -k 10 120 tools/valu_rate > $OUT/valu_rate.txt 2>&1 && echo valu done
timeout -k 10 300 tools/streamring lookup > $OUT/streamring_lookup.txt 2>&1 && echo stream done
LIB=$R/toyrenderer_amd/lib
for n in stamps stamps_nl; do
  TRHIP_LIB=$LIB/exp/$n/libtrhip.so LD_LIBRARY_PATH=$LIB/exp/$n timeout -k 10 200 python3 tools/stamps.py > $OUT/$n.txt 2> $OUT/$n.err && echo $n done
done
AB_STEPS=50 bash tools/ab.sh base nolookup -- > $OUT/ab_nolookup.txt 2>&1
cat $OUT/stamps.txt $OUT/stamps_nl.txt $OUT/ab_nolookup.txt
