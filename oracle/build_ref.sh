#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE. Builds oracle/_ref/libicar_ref.so from the
# reference sources WHERE THEY LIE under /root/reference (nothing is copied into the repo).
#
# What is compiled (unmodified, flang -O2): the reference's hot-path modules, the interface
# modules they `use` (recipe: SURVEY.md Appendix B) and the two helper modules rows T3 / W2 call
# (utilities/atm_utilities.f90, utilities/array_utilities.f90).  The NetCDF/FFTW-dependent *_obj.f90
# submodule bodies are not on the path and are not compiled; their never-called type-bound
# procedures stay unresolved (-Wl,--unresolved-symbols=ignore-all).
#
# DISCLOSURE: this image has no coarray runtime.  The reference calls this_image() only inside
# debug prints; flang lowers that to prif_this_image_no_coarray, which oracle/ref_link_stubs.f90
# answers with 1 -- the same single-image semantics as the reference CI's -fcoarray=single
# build (.github/workflows/icar-main-commit.yml).  ref_link_stubs.c provides the flang-runtime
# registration hook _FortranAAMDRegisterAllocator (no-op).  num_images() (grid_obj.f90:163 only) is
# answered with a value set by the shim so that the reference's own for_image= path yields every
# tile of an N-image decomposition in one process.  No stub does arithmetic.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
R=${ICAR_REFERENCE:-/root/reference}/src
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
OUT="$HERE/_ref"
OBJ="$OUT/obj"
[ -d "$R" ] || { echo "build_ref: $R not present (GPU box?) - using prebuilt files"; exit 0; }
mkdir -p "$OBJ"
cd "$OBJ"
FLAGS="-c -cpp -O2 -fPIC -fcoarray -w -DUSE_ASSERTIONS=.false. -I$R/physics -I$R/utilities"
# prif stub module must exist before anything using this_image() is compiled
[ ref_link_stubs.o -nt "$HERE/ref_link_stubs.f90" ] || $FC -c -O2 -fPIC -w "$HERE/ref_link_stubs.f90" -o ref_link_stubs.o 2>/dev/null
gcc -c -fPIC "$HERE/ref_link_stubs.c" -o ref_link_stubs_c.o
for f in constants/icar_constants constants/wrf_constants utilities/time_delta_obj utilities/time_h \
         main/data_structures objects/opt_types objects/options_h utilities/assertions objects/grid_h \
         objects/meta_data_h objects/variable_h objects/variable_dict_h objects/exchangeable_h \
         objects/boundary_h objects/domain_h objects/grid_obj physics/adv_mpdata physics/advect \
         physics/mp_simple physics/mp_thompson utilities/atm_utilities utilities/array_utilities physics/mp_wsm3 physics/mp_wsm6 ; do
  o=$(basename $f).o
  if [ ! -f "$o" ] || [ "$R/$f.f90" -nt "$o" ]; then
    $FC $FLAGS "$R/$f.f90" -o "$o" 2>&1 | grep -v "multi image Fortran features" || true
  fi
done
$FC $FLAGS "$HERE/ref_shim.f90" -o ref_shim.o 2>&1 | grep -v "multi image Fortran features" || true
OBJS="ref_shim.o ref_link_stubs.o ref_link_stubs_c.o \
    adv_mpdata.o advect.o mp_simple.o mp_thompson.o icar_constants.o wrf_constants.o data_structures.o \
    opt_types.o options_h.o domain_h.o grid_h.o variable_h.o variable_dict_h.o meta_data_h.o \
    exchangeable_h.o boundary_h.o time_h.o time_delta_obj.o assertions.o grid_obj.o atm_utilities.o array_utilities.o mp_wsm3.o mp_wsm6.o"
# The interface modules carry type-bound-procedure tables that point at bodies living in the
# (uncompiled, NetCDF-dependent) *_obj.f90 submodules.  They are never called on this path; bind
# each such dangling Fortran module symbol (_QM*) to address 0, which is what a static link with
# --unresolved-symbols=ignore-all does, so that the shared object can be dlopen()ed.
nm -u $OBJS | awk '$1=="U" && $2 ~ /^_QM/ {print $2}' | sort -u > undef.txt
nm --defined-only $OBJS | awk 'NF==3 {print $3}' | sort -u > def.txt
comm -23 undef.txt def.txt | sed 's/.*/-Wl,--defsym,&=0/' > defsyms.rsp
$FC -shared -o "$OUT/libicar_ref.so" $OBJS @defsyms.rsp
echo "built $OUT/libicar_ref.so"
