import csv, collections, sys, glob
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True)[0]
by=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if 'k_conv_dma' in r['Kernel_Name']:
        by[r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value'])
        by[r['Dispatch_Id']]['t0']=float(r['Start_Timestamp']); by[r['Dispatch_Id']]['t1']=float(r['End_Timestamp'])
        by[r['Dispatch_Id']]['g']=r['Grid_Size']
for k,v in list(by.items()):
    dur=(v['t1']-v['t0'])/1e3
    if dur<1000: continue
    print(k, v['g'], f"dur {dur:8.1f}us  clk={v['GRBM_GUI_ACTIVE']/8/dur/1e3:.3f} GHz  mfma_busy={v['SQ_VALU_MFMA_BUSY_CYCLES']/(v['GRBM_GUI_ACTIVE']/8*1024):.3f}")
