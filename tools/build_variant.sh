#!/bin/bash
# Build a variant of libimsegm_hip.so with extra compile flags for ONE source file, for A/B runs on the same GPU box:
#   tools/build_variant.sh pf32 slic.hip -DSLIC_PF_TX=32 -DSLIC_PF_TY=32
#   tools/build_variant.sh base                       (a copy of the current library)
# -> pyimsegm_amd/build/variants/<name>.so ; compare with  gpurun -- 'bash tools/variants_k.sh "pre_fused" base pf32'
#    (kernel averages under rocprofv3) or tools/variants.sh (assignment kernel by HIP events).  Known switches:
#    SLIC_DOT_MIN_BLOCKS (waves per SIMD of the assignment kernel), SLIC_PF_TX / SLIC_PF_TY (pre-processing tile).
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
python -m pyimsegm_amd.build > /dev/null
mkdir -p $REPO/pyimsegm_amd/build/variants
if [ $# -eq 0 ]; then cp $REPO/pyimsegm_amd/libimsegm_hip.so $REPO/pyimsegm_amd/build/variants/$NAME.so; echo "$NAME = current library"; exit 0; fi
SRC=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function"
OBJ=/tmp/variant_${NAME}_${SRC%.hip}.o
/opt/rocm/bin/hipcc $FLAGS "$@" -c $REPO/pyimsegm_amd/csrc/$SRC -o $OBJ
OBJS=$(ls $REPO/pyimsegm_amd/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/pyimsegm_amd/build/variants/$NAME.so $OBJS $OBJ
echo "built pyimsegm_amd/build/variants/$NAME.so ($SRC $*)"
