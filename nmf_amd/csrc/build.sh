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
  stale=0
  for h in "$here"/*.hpp "$here/../../include/nmf_hip.h"; do [ "$h" -nt "$o" ] && stale=1; done
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$stale" = 1 ]; then
    extra=""
    # bookkeeping kernels must reproduce the CPU oracle bit-for-bit: no a*b+c -> fma contraction there
    case "$(basename "$f")" in
      march.hip|select.hip|composite.hip|env.hip) extra="-ffp-contract=off" ;;
      # the MLP backward keeps 128 accumulator registers alive across its loop: transient MFMA results go to VGPRs directly
      # (the default picks the AGPR form for every MFMA of a 512-register kernel and copies each result out)
      brdf_mlp.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1" ;;
    esac
    # (a failed compile must not leave the previous object to be linked: remove it first, check that all exist afterwards)
    rm -f "$o"
    "$HIPCC" $FLAGS $extra -c "$f" -o "$o" &
  fi
  objs+=("$o")
done
wait
for o in "${objs[@]}"; do
  if [ ! -f "$o" ]; then echo "build.sh: $o was not produced (compile error above)" >&2; exit 1; fi
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out/libnmf_hip.so"
echo "built $out/libnmf_hip.so"

# Optional host-side fast path (csrc/host_ext.cpp: the forward wrappers of hip.py in C++, same C ABI underneath).  Plain
# g++ against the torch headers, no device code; hip.py works without it (pure-Python wrappers), so a failure is not fatal.
ext="$out/_nmf_host.so"
if [ "${NMF_BUILD_HOST_EXT:-1}" = "1" ]; then
  if [ ! -f "$ext" ] || [ "$here/host_ext.cpp" -nt "$ext" ] || [ "$here/step_core.inc" -nt "$ext" ] || [ "$here/../../include/nmf_hip.h" -nt "$ext" ]; then
    tdir="$(python3 -c 'import torch, os; print(os.path.dirname(torch.__file__))' 2>/dev/null || true)"
    pyinc="$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])' 2>/dev/null || true)"
    if [ -n "$tdir" ] && [ -n "$pyinc" ] && g++ -O2 -fPIC -shared -std=c++17 "$here/host_ext.cpp" \
         -I"$here/../../include" -I"$tdir/include" -I"$tdir/include/torch/csrc/api/include" -I"$pyinc" \
         -D_GLIBCXX_USE_CXX11_ABI=1 -DTORCH_EXTENSION_NAME=_nmf_host \
         -L"$tdir/lib" -ltorch -ltorch_cpu -lc10 -ltorch_python -L"$out" -lnmf_hip \
         -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$tdir/lib" -o "$ext.tmp" 2> "$out/host_ext.log"; then
      mv "$ext.tmp" "$ext"
      echo "built $ext"
    else
      rm -f "$ext.tmp"
      echo "warning: host extension not built (see $out/host_ext.log); hip.py falls back to its Python wrappers"
    fi
  fi
fi
