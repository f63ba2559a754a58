import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
n=int(sys.argv[2]) if len(sys.argv)>2 else 16
for r in rows[:n]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
