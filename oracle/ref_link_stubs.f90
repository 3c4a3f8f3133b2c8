!> oracle/ref_link_stubs.f90 -- TEST INFRASTRUCTURE (see build_ref.sh DISCLOSURE).
!! Single-image answer for this_image(): the only coarray-runtime (PRIF) symbol the
!! reference hot-path objects reference, and only from debug prints.
module prif
  use iso_c_binding
  implicit none
  type :: prif_team_type
     type(c_ptr) :: p = c_null_ptr
  end type
contains
  subroutine prif_this_image_no_coarray(team, this_image)
    type(prif_team_type), intent(in), optional :: team
    integer(c_int), intent(out) :: this_image
    this_image = 1
  end subroutine
end module
