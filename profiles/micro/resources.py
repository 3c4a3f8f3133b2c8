"""Register / spill / scratch table of one source's kernels: python profiles/micro/resources.py <file.hip> [extra hipcc flags ...]
(icar_amd.build.kernel_resources: the compiler's -Rpass-analysis=kernel-resource-usage remarks, one line per kernel;
tests/test_kernel_resources.py asserts the numbers the hot kernels' performance rests on)."""
import sys
sys.path.insert(0, ".")
from icar_amd import build as B
rows = B.kernel_resources(sys.argv[1], sys.argv[2:])
keys = ["TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"]
print("| kernel | " + " | ".join(keys) + " |"); print("|---" * (len(keys) + 1) + "|")
for name, r in rows.items(): print("| " + name + " | " + " | ".join(str(r.get(k, "")) for k in keys) + " |")
