import csv,sys,collections
f=sys.argv[1]; pat=sys.argv[2]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if pat in n:
        acc[n.split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()): print('   %-28s mean %.4g  n %d'%(c,sum(v)/len(v),len(v)))
