"""Round 6 (EXPERIMENTS.md G8): the fabric bytes one launch of the long-K Gram tile kernel would move if every operand line were
fetched into an XCD's L2 ONCE per round of 32 workgroups (perfect sharing inside a round, none across rounds: a chunk of one
128-row block is 4 MiB = the whole L2), for the tile orders and workgroup -> (tile, chunk) mappings that were considered.
Host arithmetic only.

    python scripts/gram_order_footprint.py
"""
import sys
def tiles(n):
    t128=-(-n//128); t256=-(-t128//2); S=-(-t128//8)
    out=[]
    for I in range(S):
        for J in range(I+1):
            for bi in range(I*4,min(I*4+4,t256)):
                for tj in range(J*8,min(J*8+8,t128)):
                    if tj<=2*bi+1: out.append((bi,tj))
    return out
def rows_of(t):
    bi,tj=t
    return {('r',2*bi),('r',2*bi+1),('r',tj)}   # 128-row blocks
def cur(n,nch,R=32):
    T=tiles(n); nt=len(T); base,rem=nt//8,nt%8
    tot=0
    for x in range(8):
        mine=base+(1 if x<rem else 0); first=x*base+min(x,rem)
        nseq=mine*nch
        for r0 in range(0,nseq,R):
            s=set()
            for seq in range(r0,min(r0+R,nseq)):
                ch=seq//mine; t=T[first+seq-ch*mine]
                for rb in rows_of(t): s.add((rb,ch))
            tot+=len(s)
    return tot*128*8192*4, nt
def new(n,nch,R=32):
    T=tiles(n); nt=len(T); units=nt*nch; tot=0
    for r0 in range(0,units,R):
        s=set()
        for u in range(r0,min(r0+R,units)):
            ch=u//nt; t=T[u%nt]
            for rb in rows_of(t): s.add((rb,ch))
        tot+=len(s)
    return tot*128*8192*4
for n,cols in ((4000,1000448),(10000,401408)):
    nch=cols//8192
    c,nt=cur(n,nch); w=new(n,nch)
    print(n,'tiles',nt,'chunks',nch,'current floor %.1f GB'%(c/1e9),'global-order floor %.1f GB'%(w/1e9), 'algorithmic %.1f'%(n*cols*4/1e9))

def order_band(n, snake=False, band=4):
    t128=-(-n//128); t256=-(-t128//2)
    out=[]
    k=0
    for b0 in range(0,t256,band):
        bis=list(range(b0,min(b0+band,t256)))
        tjs=list(range(0,min(2*bis[-1]+2,t128)))
        if snake and k%2: tjs=tjs[::-1]
        for tj in tjs:
            for bi in bis:
                if tj<=2*bi+1: out.append((bi,tj))
        k+=1
    return out
def floor_global(T,nch,R=32):
    nt=len(T); units=nt*nch; tot=0
    for r0 in range(0,units,R):
        s=set()
        for u in range(r0,min(r0+R,units)):
            ch=u//nt; t=T[u%nt]
            for rb in rows_of(t): s.add((rb,ch))
        tot+=len(s)
    return tot*128*8192*4/1e9
for n,cols in ((4000,1000448),(10000,401408),(7601,401408)):
    nch=cols//8192
    print(n, 'superblock %.1f'%floor_global(tiles(n),nch), 'band4 %.1f'%floor_global(order_band(n),nch), 'band4 snake %.1f'%floor_global(order_band(n,True),nch),
          'band3 %.1f'%floor_global(order_band(n,False,3),nch),'band5 %.1f'%floor_global(order_band(n,False,5),nch), 'band6 %.1f'%floor_global(order_band(n,False,6),nch), 'ideal %.1f'%(len(tiles(n))*nch/32*16*128*8192*4/1e9))


def order_panels(n, panel=32, band=4):
    """column panels of `panel` 128-row blocks; inside a panel band by band, inside a band column block by column block"""
    t128 = -(-n // 128); t256 = -(-t128 // 2)
    out = []
    for j0 in range(0, t128, panel):
        for b0 in range(0, t256, band):
            for tj in range(j0, min(j0 + panel, t128)):
                for bi in range(b0, min(b0 + band, t256)):
                    if tj <= 2 * bi + 1:
                        out.append((bi, tj))
    return out


def hbm_bytes(T, nch, R=32, mall_mb=256):
    """bytes that miss an LRU cache of mall_mb shared by the XCDs, runs of R units taken in sequence order"""
    from collections import OrderedDict
    lru = OrderedDict(); cap = mall_mb // 4; miss = 0      # entries of 4 MiB: one (128-row block, chunk)
    nt = len(T); units = nt * nch
    for r0 in range(0, units, R):
        s = set()
        for u in range(r0, min(r0 + R, units)):
            ch = u // nt
            for rb in rows_of(T[u % nt]): s.add((rb, ch))
        for key in s:
            if key in lru: lru.move_to_end(key)
            else:
                miss += 1; lru[key] = 1
                if len(lru) > cap: lru.popitem(last=False)
    return miss * 128 * 8192 * 4 / 1e9


if __name__ == '__main__':
    print('\nL2 floor (GB through the fabric) / HBM floor (GB missing a 256 MB LRU Infinity Cache), per launch:')
    for n, cols in ((4000, 1000448), (10000, 401408), (7601, 401408)):
        nch = cols // 8192
        for name, T in (('super-blocks', tiles(n)), ('bands of 4', order_band(n)), ('panels 32 x bands', order_panels(n)), ('panels 16 x bands', order_panels(n, 16)),('panels 24 x bands', order_panels(n, 24))):
            assert sorted(T) == sorted(tiles(n))
            print('  N = %5d  %-18s  %6.1f  /  %6.1f   (unique planes %.1f GB)' % (n, name, floor_global(T, nch), hbm_bytes(T, nch), -(-n // 128) * 128 * cols * 4 / 1e9))
