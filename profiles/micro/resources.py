"""Register / spill / scratch table of one source's kernels: python profiles/micro/resources.py <file.hip> [extra hipcc flags ...]
(the compiler's -Rpass-analysis=kernel-resource-usage remarks, one line per kernel)."""
import os, re, subprocess, sys
sys.path.insert(0, ".")
from icar_amd import build as B
src, extra = sys.argv[1], sys.argv[2:]
cmd = [B.HIPCC] + B.FLAGS + B.PER_FILE_FLAGS.get(src, []) + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.CSRC, src), "-o", "/tmp/resources.o"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "")}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
keys = ["TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"]
print("| kernel | " + " | ".join(keys) + " |"); print("|---" * (len(keys) + 1) + "|")
for r in rows: print("| " + r["name"] + " | " + " | ".join(r.get(k, "") for k in keys) + " |")
