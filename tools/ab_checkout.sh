#!/bin/bash
# A side-by-side build of an EARLIER commit under tools/ab/<name> (git-ignored; it travels to the GPU box with gpurun), for same-box
# interleaved A/B runs against this tree (tools/ab_run.sh, tools/ab_sgr.sh):   tools/ab_checkout.sh 0807f13 r3
set -e
COMMIT=$1; NAME=$2
DST=tools/ab/$NAME
rm -rf $DST; mkdir -p $DST
git archive $COMMIT image-text-retrieval_amd include oracle bench.py profiles/r02/cpu_fold | tar -x -C $DST
make -C $DST/image-text-retrieval_amd/csrc -j8 > $DST/build.log 2>&1 || { tail -20 $DST/build.log; exit 1; }
rm -rf $DST/image-text-retrieval_amd/csrc/build
ls -la $DST/image-text-retrieval_amd/itr_amd/libitr_hip.so
