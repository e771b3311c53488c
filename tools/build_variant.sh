# tools/build_variant.sh NAME SOURCE.hip "-DFLAG ..." : variants/NAME.so = the current library with SOURCE rebuilt under extra flags
set -e
REPO=$(cd $(dirname $0)/.. && pwd)
NAME=$1; SRC=$2; FLAGS=$3
mkdir -p $REPO/variants /tmp/bl_variants
OBJ=/tmp/bl_variants/$NAME.o
hipcc -c $REPO/blacklight_amd/csrc/$SRC -o $OBJ -std=c++17 -O3 -ffp-contract=off -fPIC -fvisibility=hidden -I$REPO/include -I$REPO/blacklight_amd/csrc --offload-arch=gfx950 -mllvm -disable-machine-licm $FLAGS
OTHERS=$(ls $REPO/blacklight_amd/csrc/_obj/*.o | grep -v "/${SRC%.*}.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o $REPO/variants/$NAME.so $OBJ $OTHERS
echo built variants/$NAME.so
