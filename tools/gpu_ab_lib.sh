# One-box A/B of two BUILDS of the library: bash tools/gpu_ab_lib.sh OUT build_ab/old.so [more.so ...]; variant 1 is the in-tree
# library, variant k+1 the k-th argument (HRF_LIB_PATH).  Build the other version with tools/build_ab.sh NAME (library of the
# committed tree -> build_ab/NAME.so, git-ignored).  EXTRA="HRF_LANES=0" adds environment to every variant: with one lane the
# step is the SERIAL sum of all launches, the sensitive metric for a change of a kernel's latency.
set -u
cd "${GRAFT_REPO_ROOT:?}"
O=$1; shift
X=${EXTRA:-}
args=("${X:--}")
for so in "$@"; do args+=("$X HRF_LIB_PATH=$PWD/$so"); done
bash tools/gpu_ab.sh $O "${args[@]}"
