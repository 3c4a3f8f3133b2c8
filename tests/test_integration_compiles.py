"""the binding INTEGRATION.md describes, type-checked against the REFERENCE's own domain_t / options_t.

Build container only (skips where /root/reference is absent: nothing of the reference travels).  tests/support/integration_glue.f90
`use`s the reference's domain_interface / options_interface / grid_interface / options_types / time_object / icar_constants -- the
.mod files oracle/build_ref.sh compiles from /root/reference/src unmodified -- next to icar_hip, and holds the statements of
INTEGRATION.md's ```fortran blocks (sections 2 - 4).  flang -fsyntax-only must accept it; every statement of those blocks must be in
it; and a glue with a member the reference does not have must be rejected (the check is not vacuous)."""
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ICAR_REFERENCE", "/root/reference")
FLANG = os.environ.get("FLANG", "/opt/rocm/lib/llvm/bin/flang")
GLUE = os.path.join(ROOT, "tests", "support", "integration_glue.f90")
REFMOD = os.path.join(ROOT, "oracle", "_ref", "obj")
HIPMOD = os.path.join(ROOT, "icar_amd", "lib")

pytestmark = pytest.mark.skipif(not (os.path.isdir(os.path.join(REF, "src")) and os.path.exists(FLANG)),
                                reason="needs /root/reference and flang (build container only)")


@pytest.fixture(scope="module")
def modules():
    if not os.path.exists(os.path.join(REFMOD, "domain_interface.mod")):
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")])
    if not os.path.exists(os.path.join(HIPMOD, "icar_hip.mod")):
        from icar_amd import build as B
        B.build_fortran_host()
    assert os.path.exists(os.path.join(REFMOD, "domain_interface.mod")) and os.path.exists(os.path.join(HIPMOD, "icar_hip.mod"))


def syntax_check(path):
    r = subprocess.run([FLANG, "-fsyntax-only", "-cpp", "-fcoarray", "-I" + REFMOD, "-I" + HIPMOD, path], capture_output=True, text=True)
    return r.returncode, "\n".join(l for l in r.stderr.splitlines() if "multi image" not in l)


def statements(text):
    """Fortran statements of a free-form text: comments dropped, continuation lines joined, split at ';', blanks collapsed,
    this%hip / domain%hip -> hip (the one edit the glue cannot make: the member of domain_t)"""
    lines = []
    for raw in text.splitlines():
        line = re.sub(r"!.*$", "", raw).strip()          # (no '!' inside character literals in these snippets, except the error stops:
        if raw.strip().startswith("!") or not line:      #  those carry none)
            continue
        lines.append(line)
    joined, cur = [], ""
    for l in lines:
        cont = l.endswith("&")
        cur += " " + l.rstrip("&").lstrip("&").strip()
        if not cont:
            joined.append(cur); cur = ""
    out = []
    for l in joined:
        for st in l.split(";"):
            st = re.sub(r"\s+", "", st).lower().replace("this%hip", "hip").replace("domain%hip", "hip")
            if st:
                out.append(st)
    return out


def test_glue_compiles_against_the_reference_modules(modules):
    rc, err = syntax_check(GLUE)
    assert rc == 0, err


def test_every_documented_statement_is_in_the_glue(modules):
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```fortran\n(.*?)```", doc, flags=re.S)
    assert len(blocks) >= 8
    have = set(statements(open(GLUE).read()))
    missing = [st for b in blocks for st in statements(b) if st not in have]
    assert not missing, "INTEGRATION.md statements that the compile-checked glue does not contain:\n  " + "\n  ".join(missing)


@pytest.mark.parametrize("old,new", [("this%jacobian_u)", "this%jacobian_x)"),                                   # no such member of domain_t
                                     ("options%adv_options%mpdata_order;", "options%adv_options%mpdata_ordre;"),   # nor of adv_options_type
                                     ("call hip_set_model_time(hip, t0)", "call hip_set_model_time(hip, domain%model_time%seconds())")])  # real128
def test_a_wrong_binding_is_rejected(modules, tmp_path, old, new):
    src = open(GLUE).read()
    assert old in src
    p = tmp_path / "bad_glue.f90"
    p.write_text(src.replace(old, new, 1))
    rc, err = syntax_check(str(p))
    assert rc != 0 and "error" in err.lower()
