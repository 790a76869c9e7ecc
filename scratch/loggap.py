import re, sys
lines = open(sys.argv[1], errors="replace").read().split("\n")
skip = ("hipGetDevice", "hipSetDevice", "hipDriverGetVersion", "hipStreamGetCaptureInfo", "hipStreamIsCapturing", "hipGetLastError", "hipPeekAtLastError")
api = []
for i, l in enumerate(lines):
    m = re.search(r":\s*(\d{9,}) us:\s*(.*)", l)
    if not m:
        continue
    t, rest = int(m.group(1)), m.group(2)
    rest = re.sub(r"\x1b\[[0-9;]*m", "", rest).strip()
    if re.match(r"hip\w+ \(", rest) and not rest.startswith(skip):
        api.append((t, rest))
    elif "Returned" in rest and "hipSuccess" not in rest:
        api.append((t, rest))
    elif "Host active wait" in rest or "Graph shader name" in rest and False:
        api.append((t, rest))
out = []
for k in range(1, len(api)):
    if api[k][0] - api[k - 1][0] > 1000000:
        out.append("GAP %d us" % (api[k][0] - api[k - 1][0]))
        for t, r in api[max(0, k - 500):k + 30]:
            out.append("%d %s" % (t, r[:200]))
open(sys.argv[2], "w").write("\n".join(out))
print(len(lines), len(api), len(out))
