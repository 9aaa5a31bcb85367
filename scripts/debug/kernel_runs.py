import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"][:60] for r in rows]
n = len(names)
seg = names[-130:]
prev = None; cnt = 0
for x in seg + [None]:
    if x == prev: cnt += 1
    else:
        if prev is not None: print(cnt, prev)
        prev, cnt = x, 1
