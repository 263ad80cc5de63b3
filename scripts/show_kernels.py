import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    k = r["kernel"]; short = k.split("::")[-1].split("(")[0][:64]
    print("%-66s calls=%4s avg=%10s grid=%s vgpr=%s lds=%s" % (short, r["calls"], r["avg_us"], r["grid_x"], r["vgpr"], r["lds_bytes"]))
