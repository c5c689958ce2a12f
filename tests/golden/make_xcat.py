"""Re-type the reference's test data (test/testdata.h makeXdata / makeOdata)
as a JSON data fixture.  Run in the build container only (reads
/root/reference); the committed xcat.json is what travels.

    python tests/golden/make_xcat.py
"""
import json
import re
from pathlib import Path

src = Path("/root/reference/test/testdata.h").read_text()


def blocks(fn_name):
    body = src[src.index("void " + fn_name):]
    body = body[: body.index("\n}\n")]
    out = []
    for m in re.finditer(r"<<(.*?);", body, flags=re.S):
        nums = [float(t) for t in re.findall(r"-?\d+\.\d+", m.group(1))]
        assert len(nums) % 2 == 0
        out.append([[nums[2 * i], nums[2 * i + 1]] for i in range(len(nums) // 2)])
    return out


X = blocks("makeXdata")
O = blocks("makeOdata")
assert len(X) == 12 and all(len(g) == 10 for g in X), [len(g) for g in X]
assert len(O) == 2 and all(len(g) == 6 for g in O)
Path(__file__).with_name("xcat.json").write_text(
    json.dumps({"source": "test/testdata.h:30-220 (data values only)", "X": X, "O": O})
)
print("groups", len(X), "rows", sum(len(g) for g in X))
