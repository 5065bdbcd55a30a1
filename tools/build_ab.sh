# Build the library of the COMMITTED tree into build_ab/$1.so (for tools/gpu_ab_lib.sh), leaving the working tree as it was.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
git stash -q
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1 || { git stash pop -q; exit 1; }
cp hrfuser_amd/libhrfuser_hip.so build_ab/$1.so
git stash pop -q
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
ls -la build_ab/$1.so
