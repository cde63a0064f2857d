#!/bin/bash
# A/B builds of libspecmi.so with compile-time knobs of the persistent walker (experiments only):
#   scripts/build_variants.sh name "-DFOO=1 -DBAR=2" ...   ->  spec_amd/lib/variants/libspecmi_<name>.so  (select with SPECMI_LIB)
set -e
cd "$(dirname "$0")/.."
mkdir -p spec_amd/lib/variants
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  d=/tmp/specmi_variant_$name; mkdir -p $d
  pids=()
  for f in spec_amd/csrc/*.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $defs -c $f -o $d/$(basename $f .hip).o & pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o spec_amd/lib/variants/libspecmi_$name.so $d/*.o
  echo built spec_amd/lib/variants/libspecmi_$name.so
done
