#!/bin/bash
# Same-box A/B of whole trees: builds revision <rev> (default HEAD) of the package into _ab_base/ (git-ignored, travels to the
# GPU box with the snapshot), so that one gpurun call can interleave `python _ab_base/bench.py` with `python bench.py`.
# Run in the build container: bash tools/ab_tree.sh [rev]; remove _ab_base/ when done.
set -e
REV=${1:-HEAD}
cd "$(git rev-parse --show-toplevel)"
rm -rf _ab_base && mkdir _ab_base
git archive "$REV" unet_amd bench.py oracle include profiles/r02_pmc_traffic.json | tar -x -C _ab_base
make -C _ab_base/unet_amd/csrc -j8 > /dev/null
echo "built $(git rev-parse --short "$REV") into _ab_base/"
