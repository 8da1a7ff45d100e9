import csv,sys,glob
f=glob.glob('/tmp/prof/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'szg::' in r['Name']:
        print(r['Name'].split('(')[0][-30:], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1))
