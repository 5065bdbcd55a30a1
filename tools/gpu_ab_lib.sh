# One-box A/B of two BUILDS of the library: bash tools/gpu_ab_lib.sh OUT build_ab/old.so [more.so ...]; variant 1 is the in-tree
# library, variant k+1 the k-th argument (HRF_LIB_PATH).  Build the other version with `git stash; python -c "import
# __graft_entry__ as g; g.build()"; cp hrfuser_amd/libhrfuser_hip.so build_ab/old.so; git stash pop` (build_ab/ is git-ignored).
set -u
cd "${GRAFT_REPO_ROOT:?}"
O=$1; shift
args=("-")
for so in "$@"; do args+=("HRF_LIB_PATH=$PWD/$so"); done
bash tools/gpu_ab.sh $O "${args[@]}"
