# Build the library of the WORKING tree with extra compiler flags into build_ab/$1.so (for tools/gpu_ab_lib.sh), in a scratch
# copy: the in-tree library and objects are untouched.   bash tools/build_variant.sh NAME "-DFOO=1 -DBAR=2"
set -e
cd "$(dirname "$0")/.."
NAME=${1:?name}; FLAGS=${2:-}
mkdir -p build_ab
TMP=$(mktemp -d /tmp/hrf_build_var.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/hrfuser_amd/csrc" "$TMP/include"
cp hrfuser_amd/*.py "$TMP/hrfuser_amd/"; cp hrfuser_amd/csrc/*.hip hrfuser_amd/csrc/*.h "$TMP/hrfuser_amd/csrc/"; cp include/*.h "$TMP/include/"
(cd "$TMP" && HRF_EXTRA_FLAGS="$FLAGS" python -m hrfuser_amd.build_ext --force > build.log 2>&1) || { tail -20 "$TMP/build.log"; exit 1; }
cp "$TMP/hrfuser_amd/libhrfuser_hip.so" "build_ab/$NAME.so"
cp "$TMP/hrfuser_amd/kernel_resources.json" "build_ab/$NAME.resources.json"
ls -la "build_ab/$NAME.so"
