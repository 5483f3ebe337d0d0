import csv,sys
rows=list(csv.DictReader(open("gpurun_out/%s/reg_kernel_stats.csv" % sys.argv[1])))
steps=158
for r in rows[:60]:
    if "dpd::" in r["Name"]: print("%7.1f us/step %5.1f/step avg %6.2f %s" % (float(r["TotalDurationNs"])/1e3/steps, int(r["Calls"])/steps, float(r["AverageNs"])/1e3, r["Name"][:90]))
print(sum(float(r["TotalDurationNs"]) for r in rows)/1e3/steps)
