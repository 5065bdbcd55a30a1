# Timing variants of csrc/wgrad3x_engine.hip (-DW3_VARIANT=n) linked against the in-tree objects into build_ab/w3x_v<n>.so
set -e
cd "$(dirname "$0")/.."
python -m hrfuser_amd.build_ext > /dev/null
mkdir -p build_ab
C=hrfuser_amd/csrc
OBJS=$(ls $C/*.o | grep -v wgrad3x_engine.o)
for v in ${1:-1 2 3}; do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DW3_VARIANT=$v -c $C/wgrad3x_engine.hip -o /tmp/w3x_v$v.o &
done
wait
for v in ${1:-1 2 3}; do
  g++ -shared -o build_ab/w3x_v$v.so $OBJS /tmp/w3x_v$v.o -L/usr/local/lib/python3.10/dist-packages/torch/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
done
