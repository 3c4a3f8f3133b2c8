!> oracle/ref_link_stubs.f90 -- TEST INFRASTRUCTURE (see build_ref.sh DISCLOSURE).
!! Single-image answer for this_image(): the only coarray-runtime (PRIF) symbol the
!! reference hot-path objects reference, and only from debug prints.
!! num_images() (used by grid_obj.f90:163 to size the decomposition) answers with a value the
!! shim sets (ref_set_num_images), so that one process can ask the reference for the tile of
!! any image of an N-image run through its own for_image= argument -- what
!! src/tests/test_caf_other_image_grids.f90 does across real images.
module prif
  use iso_c_binding
  implicit none
  type :: prif_team_type
     type(c_ptr) :: p = c_null_ptr
  end type
  integer(c_int), save :: stub_num_images = 1
contains
  subroutine prif_num_images(num_images)
    integer(c_int), intent(out) :: num_images
    num_images = stub_num_images
  end subroutine
  subroutine prif_this_image_no_coarray(team, this_image)
    type(prif_team_type), intent(in), optional :: team
    integer(c_int), intent(out) :: this_image
    this_image = 1
  end subroutine
end module
