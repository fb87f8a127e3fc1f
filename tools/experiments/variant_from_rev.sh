#!/bin/bash
# Build build/variants/libcdml_<tag>.so from the CURRENT csrc/ with the named files taken from a git revision
# instead (a same-box A/B of one kernel: CDML_LIB_PATH=build/variants/libcdml_<tag>.so python tools/...).
# usage: tools/experiments/variant_from_rev.sh TAG REV file.hip [file2.hip ...]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/collaborative-deep-metric-learning_amd/csrc
TAG=$1; REV=$2; shift 2
d=$(mktemp -d)
cp $CSRC/*.hip $CSRC/*.h $d/
for f in "$@"; do git -C $ROOT show $REV:collaborative-deep-metric-learning_amd/csrc/$f > $d/$f; done
mkdir -p $ROOT/build/variants
# (the binding looks every declared symbol up: a variant carries an id too -- CDML_LIB_PATH libraries are exempt from the match)
echo 'extern "C" const char *cdml_build_id(void) { return "CDML_BUILD_ID=variant-'$TAG'"; }' > $d/build_id_variant.hip
(cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-array-bounds -I$CSRC -o $ROOT/build/variants/libcdml_$TAG.so *.hip 2>/dev/null)
rm -rf $d
echo built build/variants/libcdml_$TAG.so
