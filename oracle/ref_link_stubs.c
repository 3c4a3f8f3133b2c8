/* oracle/ref_link_stubs.c -- TEST INFRASTRUCTURE (see build_ref.sh DISCLOSURE).
 * flang-compiled objects that hold allocatable derived types reference this AMD flang
 * runtime registration hook; the static runtime in this image does not define it. No-op. */
void _FortranAAMDRegisterAllocator(void) {}
