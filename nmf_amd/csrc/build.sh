#!/usr/bin/env bash
# Builds libnmf_hip.so (gfx950 only) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-fast-math -Wall -Wno-unused-function"
objs=()
for f in "$here"/*.hip; do
  o="$out/$(basename "${f%.hip}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$here/common.hpp" -nt "$o" ] || [ "$here/../../include/nmf_hip.h" -nt "$o" ]; then
    extra=""
    # bookkeeping kernels must reproduce the CPU oracle bit-for-bit: no a*b+c -> fma contraction there
    case "$(basename "$f")" in march.hip|select.hip|composite.hip|env.hip) extra="-ffp-contract=off" ;; esac
    "$HIPCC" $FLAGS $extra -c "$f" -o "$o" &
  fi
  objs+=("$o")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out/libnmf_hip.so"
echo "built $out/libnmf_hip.so"
