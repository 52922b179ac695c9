"""join the parts of an accuracy record (gpurun_out/r06_accuracy_teacher_part{1,2a,2b,3}.txt, one gpurun call each: tools/accuracy_evidence.sh) into
profiles/r06_accuracy_teacher.txt; refuses parts taken with different source hashes.   usage: python tools/accuracy_join.py [out] part..."""
import sys
out, parts = sys.argv[1], sys.argv[2:]
hashes, body = set(), []
for p in parts:
    lines = open(p).read().splitlines()
    assert lines and "csrc_sha256_16=" in lines[0], p
    hashes.add(lines[0].split("csrc_sha256_16=")[1].split()[0])
    body += lines[1:]
assert len(hashes) == 1, f"parts were taken with different trees: {hashes}"
open(out, "w").write("\n".join([open(parts[0]).readline().rstrip("\n")] + body) + "\n")
print(out, len(body), "lines, source hash", hashes.pop())
