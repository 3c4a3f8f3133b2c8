"""Generates tests/golden/thompson_cache_sha256.json: SHA-256 digests and sizes of the three lookup-table cache files
the COMPILED REFERENCE writes (thompson_init in an empty directory, default mp_options).  Data only (digests)."""
import hashlib
import json
import os
import sys
import tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref

d = tempfile.mkdtemp(prefix="thompson_cache_")
ref.thompson_init(workdir=d)                      # the reference integrates the tables and writes the three files here
out = {}
for f in ("qr_acr_qg_mpt.dat", "qr_acr_qs_mpt.dat", "freezeH2O_mpt.dat"):
    p = os.path.join(d, f)
    out[f] = {"sha256": hashlib.sha256(open(p, "rb").read()).hexdigest(), "bytes": os.path.getsize(p)}
json.dump({"source": "files written by the compiled reference (oracle/_ref: thompson_init, default mp_options); "
                     "tests/golden/make_golden_thompson_cache.py", "files": out},
          open(os.path.join(HERE, "thompson_cache_sha256.json"), "w"), indent=1)
print(out)
