# Timing variants of csrc/conv3x_engine.hip (-DX_VARIANT=n: a part of the kernel compiled out) linked against the in-tree objects
# into build_ab/c3x_v<n>.so; run with HRF_LIB_PATH=build_ab/c3x_v<n>.so python tools/bench_conv3x.py.  bash tools/c3x_variants.sh "1 2 3"
set -e
cd "$(dirname "$0")/.."
python -m hrfuser_amd.build_ext > /dev/null
mkdir -p build_ab
C=hrfuser_amd/csrc
OBJS=$(ls $C/*.o | grep -v conv3x_engine.o)
for v in ${1:-1 2 3 4 5 6}; do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off ${EXTRA:-} -DX_VARIANT=$v -c $C/conv3x_engine.hip -o /tmp/c3x_v$v.o &
done
wait
for v in ${1:-1 2 3 4 5 6}; do
  g++ -shared -o build_ab/c3x_v$v.so $OBJS /tmp/c3x_v$v.o -L/usr/local/lib/python3.10/dist-packages/torch/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
done
ls -la build_ab/
