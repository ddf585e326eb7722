import os
os.environ["X3_STAMP_WGS"]="1024"
exec(open("tools/dbg_stamps.py").read().split("a = out.reshape")[0])
a = out.reshape(-1, 8).astype(np.float64)
names="analyze,scan,B1wait+total,dma+zero+cbar|lookback,emit,dmawait+B3wait,crc,B4wait+copy".split(",")
for who,sl in (("compute wave0",a[0::2]),("helper wave8",a[1::2])):
    sl=sl[sl.sum(axis=1)>0]
    print(who, "WGs", len(sl), "total", sl.sum(axis=1).mean())
    for k in range(8): print("   %-18s mean %10.0f" % (names[k], sl[:,k].mean()))
