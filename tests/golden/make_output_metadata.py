"""tests/golden/make_output_metadata.py -> tests/golden/output_metadata.json   (build container only: reads /root/reference).

What an ICAR output / restart file looks like, DERIVED from the reference's sources instead of hand-copied:
  * src/io/default_output_metadata.f90: for every kVARS entry the file variable name, the dimension-name list (Fortran order, as
    written there), unlimited_dim and the attribute list, in order;
  * src/io/output_obj.f90: the file format flag of nf90_create (:54), the time variable (type, attributes, :380-399), the global
    attributes (add_global_attributes, :286-330), the data type rule (kREAL -> NF90_REAL, kDOUBLE -> NF90_DOUBLE, :500-507) and the
    memory -> file index order reshape(order=[1,3,2]) (:423);
  * src/utilities/time_obj.f90:570-578: the format of the time units string.
tests/test_output_netcdf.py holds icar_amd/output.py:METADATA and the header of a file it writes to this fixture.  The fixture is
data (names, dimension lists, attribute strings), not source text."""
import json, os, re, sys
REF = os.environ.get("ICAR_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def strip_comment(line):
    out, q = "", None
    for ch in line:
        if q:
            out += ch
            if ch == q: q = None
        elif ch in "\"'":
            q = ch; out += ch
        elif ch == "!":
            break
        else:
            out += ch
    return out.rstrip()


def logical_lines(path):
    """free-form source -> statements (comments dropped, & continuations joined)"""
    cur = ""
    for raw in open(path, errors="replace"):
        l = strip_comment(raw.rstrip("\n")).strip()
        if not l:
            continue
        if l.endswith("&"):
            cur += l[:-1].rstrip() + " "
            continue
        cur += l.lstrip("&") if cur else l
        yield cur
        cur = ""


def parse_metadata(path):
    dims, out, cur = {}, {}, None
    for st in logical_lines(path):
        m = re.match(r"character\(len=\d+\)\s*::\s*(\w+)\(\d+\)\s*=\s*\[character\(len=\d+\)\s*::\s*(.*)\]", st, re.I)
        if m:
            dims[m.group(1).lower()] = re.findall(r'"([^"]*)"', m.group(2)); continue
        m = re.match(r"associate\s*\(\s*var\s*=>\s*var_meta\(\s*kVARS%(\w+)\s*\)\s*\)", st, re.I)
        if m:
            cur = {"name": None, "dimensions": None, "unlimited_dim": False, "attributes": []}; out[m.group(1)] = cur; continue
        if cur is None:
            continue
        if re.match(r"end\s*associate", st, re.I):
            cur = None; continue
        m = re.match(r"var%name\s*=\s*\"([^\"]*)\"", st, re.I)
        if m: cur["name"] = m.group(1); continue
        m = re.match(r"var%dimensions\s*=\s*(\w+)", st, re.I)
        if m: cur["dimensions"] = dims[m.group(1).lower()]; continue
        m = re.match(r"var%unlimited_dim\s*=\s*\.(true|false)\.", st, re.I)
        if m: cur["unlimited_dim"] = m.group(1).lower() == "true"; continue
        if re.match(r"var%attributes\s*=", st, re.I):
            cur["attributes"] = [[k, v] for k, v in re.findall(r'attribute_t\(\s*"([^"]*)"\s*,\s*"([^"]*)"\s*\)', st)]
    return {k: v for k, v in out.items() if v["name"] is not None}


def parse_output_obj(path):
    src = list(logical_lines(path))
    text = "\n".join(src)
    create = re.search(r"nf90_create\(\s*filename\s*,\s*(\w+)", text).group(1)
    glob = re.findall(r'nf90_put_att\(\s*(?:ncid|this%ncfile_id)\s*,\s*NF90_GLOBAL\s*,\s*"(\w+)"\s*,\s*(.*?)\)\s*(?:,\s*(?:trim\(err\)|"[^"]*")\s*\))?$', text, re.M)
    g = []
    for k, v in glob:
        lit = re.match(r'^"([^"]*)"$', v.strip())
        entry = [k, lit.group(1) if lit else None]              # None: computed at run time (history, git, image)
        if entry not in g and not (entry[1] is None and any(e[0] == k for e in g)):
            g.append(entry)
    tm = re.search(r'nf90_def_var\(this%ncfile_id,\s*var%name,\s*(NF90_\w+),\s*var%dim_ids\(1\)', text).group(1)
    tatts = re.findall(r'nf90_put_att\(this%ncfile_id,\s*var%var_id,\s*"(\w+)"\s*,\s*(.*?)\)\)', text)
    order = re.search(r"reshape\(var%data_3d,\s*shape=dim_3d,\s*order=\[([\d,]+)\]\)", text).group(1)
    dtypes = dict(re.findall(r"var%dtype\s*==\s*(k\w+)\)\s*then\s*\n\s*call check\(\s*nf90_def_var\(this%ncfile_id,\s*var%name,\s*(NF90_\w+)", text))
    return {"nf90_create_mode": create, "global_attributes": g, "time_type": tm,
            "time_attributes": [[k, (re.match(r'^"([^"]*)"$', v.strip()).group(1) if re.match(r'^"([^"]*)"$', v.strip()) else None)] for k, v in tatts],
            "reshape_order_3d": [int(x) for x in order.split(",")], "dtype_rule": dtypes}


def parse_time_units(path):
    text = "\n".join(logical_lines(path))
    m = re.search(r"write\(units,\s*'\((.*?)\)'\)", text)
    return m.group(1)


def main():
    io = os.path.join(REF, "src", "io")
    meta = parse_metadata(os.path.join(io, "default_output_metadata.f90"))
    res = {"source": "src/io/default_output_metadata.f90, src/io/output_obj.f90, src/utilities/time_obj.f90 of the reference (parsed, not copied)",
           "output_obj": parse_output_obj(os.path.join(io, "output_obj.f90")),
           "time_units_format": parse_time_units(os.path.join(REF, "src", "utilities", "time_obj.f90")),
           "variables": meta}
    json.dump(res, open(os.path.join(HERE, "output_metadata.json"), "w"), indent=1, sort_keys=True)
    print(len(meta), "variables;", res["output_obj"], res["time_units_format"])


if __name__ == "__main__":
    main()
