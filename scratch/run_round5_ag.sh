#!/bin/bash
# round 5, call ag: what about S from memory slows the REST of an fp32 step down?  fp32 RoBERTa-base, Gaussian, arms "fused everywhere"
# and "S from memory on the 768-wide layers only", one process per build of the sketch (same lease, back to back, production first and last):
#   noxcd         the from-memory product kernel in the grid's own tile order
#   ahead2        fragment loads 2 steps ahead instead of 4
#   nofragkernel  no fragment launch (the product kernel reads stale bytes: wrong results, same instruction stream)
#   fragonly      the fragment launch, then the FUSED product kernel (fragments not read)
#   storent / loadnt / fragfirst   non-temporal stores of the fragments / loads of them / fragment launch in front of the conversion pass
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
T=$(date +%H%M%S)
OUT=gpurun_out/r05ag_variants_$T.txt
bash scratch/box_fingerprint.sh | grep -i "vbios_version\|smc\|MEC firm" | head -4 > $OUT
for v in ${VARIANTS:-production noxcd ahead2 nofragkernel fragonly production}; do
    echo "== $v" >> $OUT
    if [ $v = production ]; then unset FEWBIT_HIP_LIB; else export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_$v.so; fi
    ARMS=0,2 timeout 300 python scratch/roberta_ab_width.py 2 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
