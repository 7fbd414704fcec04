#!/bin/bash
# Build a copy of THIS tree under tools/ab/<name> (git-ignored; it travels to the GPU box with gpurun) with extra compiler flags, for
# same-box A/B runs of kernel experiments (always with -DITR_EXPERIMENT: the environment switches of csrc/ exist in these builds only):   tools/ab_build.sh exp1 -DSF_EXP_SAME_BLOCK=1    ->  python3 tools/ab/exp1/bench.py ...
set -e
NAME=$1; shift
DST=tools/ab/$NAME
rm -rf $DST; mkdir -p $DST
cp -r image-text-retrieval_amd include oracle bench.py $DST/
mkdir -p $DST/profiles; cp -r profiles/r02 $DST/profiles/ 2>/dev/null || true
rm -rf $DST/image-text-retrieval_amd/csrc/build $DST/image-text-retrieval_amd/itr_amd/libitr_hip.so
make -C $DST/image-text-retrieval_amd/csrc -j8 EXTRA="-DITR_EXPERIMENT $*" > $DST/build.log 2>&1 || { tail -20 $DST/build.log; exit 1; }
rm -rf $DST/image-text-retrieval_amd/csrc/build
ls -la $DST/image-text-retrieval_amd/itr_amd/libitr_hip.so
