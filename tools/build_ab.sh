# Build the library of a COMMITTED tree (default HEAD) into build_ab/$1.so (for tools/gpu_ab_lib.sh) without touching the
# working tree: the commit is exported into a scratch directory (git archive) and built there.
#   bash tools/build_ab.sh NAME [COMMIT]
set -e
cd "$(dirname "$0")/.."
NAME=${1:?name of the build, e.g. old}
REV=${2:-HEAD}
mkdir -p build_ab
TMP=$(mktemp -d /tmp/hrf_build_ab.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
git archive "$REV" hrfuser_amd include | tar -x -C "$TMP"
(cd "$TMP" && python -m hrfuser_amd.build_ext --force > build.log 2>&1) || { tail -20 "$TMP/build.log"; exit 1; }
cp "$TMP/hrfuser_amd/libhrfuser_hip.so" "build_ab/$NAME.so"
ls -la "build_ab/$NAME.so"
