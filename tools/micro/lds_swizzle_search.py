rgroups=[list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32))]
rgroups += [[x+32 for x in g] for g in rgroups]
wgroups=[list(range(8*a,8*a+8)) for a in range(8)]
def ok(s):
    for k in range(4):
        for g in rgroups:
            qs=set()
            for j in g:
                jj=j&15
                a=(j>>4)*1024 + jj*64 + ((k^s[jj])*16)
                qs.add((a//16)%16)
            if len(qs)!=16: return False
        for g in wgroups:
            qs=set()
            for j in g:
                jj=j&15
                a=(j>>4)*1024 + jj*64 + ((k^s[jj])*16)
                qs.add((a//16)%8)
            if len(qs)!=8: return False
    return True
found=[]
for m0 in range(16):
    for m1 in range(16):
        s=[(bin(jj&m0).count('1')&1) | ((bin(jj&m1).count('1')&1)<<1) for jj in range(16)]
        if ok(s): found.append((m0,m1,s))
print(len(found))
for f in found[:10]: print(f)
