import csv,re,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[]
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']; n=re.sub(r'rlnamd::','',n); m=re.match(r'(?:void )?([A-Za-z0-9_]+)',n); t=m.group(1)
    if 'k_msm29' in n: t+='<G2>' if ('G2Acc' in n) else '<G1>'
    if 'k_sum' in n: t+='<Fq2>' if ('Fq2' in n or 'Fp2' in n) else '<Fq>'
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),t,r.get('Queue_Id',''),r.get('Stream_Id','')))
rows.sort()
starts=[i for i,r in enumerate(rows) if r[2]=='k_witness_lanes']
for i0 in starts[-2:]:
    t0=rows[i0][0]; j=i0-2
    while j<len(rows) and (j<=i0 or rows[j][2]!='k_witness_lanes'):
        s,e,n,q,st=rows[j]
        print("%-26s %8.3f %7.3f q%s s%s"%(n,(s-t0)/1e6,(e-s)/1e6,q,st)); j+=1
    print('----')
