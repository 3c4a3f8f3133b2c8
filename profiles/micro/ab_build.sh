#!/bin/bash
# profiles/micro/ab_build.sh <src-stem> <name> [extra hipcc flags...] -- A/B variant of libicar_hip.so with ONE object rebuilt
# with extra flags (e.g. a -D macro of an experiment patched into the source for the occasion): icar_amd/lib/ab/lib_<name>.so.
# profiles/micro/ab_libs.sh then runs bench.py once per variant through ICAR_HIP_LIB.  The product build has no such macros.
set -e
cd "$(dirname "$0")/../.."
stem=$1; name=$2; shift 2
mkdir -p icar_amd/lib/ab
extra=""
[ "$stem" = "mpdata" ] && extra="-fno-honor-nans -ffp-contract=fast"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $extra -Iinclude -Iicar_amd/csrc "$@" -c icar_amd/csrc/$stem.hip -o icar_amd/lib/ab/${stem}_$name.o 2>&1 | grep -E "error" || true
objs=$(ls icar_amd/lib/*.o | grep -v "/$stem.o\|icar_hip_mod.o\|demo")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o icar_amd/lib/ab/lib_$name.so $objs icar_amd/lib/ab/${stem}_$name.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
echo "built icar_amd/lib/ab/lib_$name.so"
