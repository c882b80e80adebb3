#!/bin/bash
# Build a tuning / debug variant of the library next to the production one:
#   tools/build_variant.sh <tag> [extra hipcc flags...]   ->  cosmology-model-fit_amd/libcosmofit_hip_<tag>.so
# Use it with COSMOFIT_LIB=<path> (cosmology-model-fit_amd/_lib.py).  Variants are never loaded by default.
set -e
tag=$1; shift
cd "$(dirname "$0")/../cosmology-model-fit_amd/csrc"
make OUT=../libcosmofit_hip_${tag}.so EXTRA="$*"
