import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"][:70] for r in rows]
# last forward: from the last front_bf16_kernel to the end
idx = max(i for i, n in enumerate(names) if "front_bf16" in n)
for n in names[idx - 3: idx + 22]:
    print(n)
