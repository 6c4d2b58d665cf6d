import sys
blk=int(sys.argv[2])
for line in open(sys.argv[1]):
    f=line.split()
    if int(f[0])!=blk: continue
    ev=[tuple(int(x) for x in e.split(':')) for e in f[2:]]
    if not ev: print(f[1],'-'); continue
    t0=ev[0][0]
    # print deltas for events 40..80
    seg=ev[40:72]
    print(f[1], ' '.join(f"{b}+{t1-t0_}" for (t0_,a),(t1,b) in zip(seg,seg[1:])))
