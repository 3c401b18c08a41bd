# For every kernel of an AMDGPU assembly file (hipcc -S --cuda-device-only): registers read inside a loop while not definitely assigned on every
# path from the kernel entry (a forward must-be-defined analysis over the CFG of the assembly), reads by plain copies (v_mov*: the phi copies
# of loop-carried values -- the shape of round 6 defect, profiles/r06s_upsample_nondeterminism.md) listed apart from reads by other
# instructions.  Conservative: paths the program never takes count (a value set and used under the same condition is reported).
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S rvdd-release_amd/csrc/conv3x3h.hip -o /tmp/c.s && python tools/undefined_in_loops.py /tmp/c.s
import re,sys,collections,subprocess
src=sys.argv[1]
text=open(src).read().split('\n')
# split into functions
funcs=[]; cur=None
for i,l in enumerate(text):
    m=re.match(r'^(_Z\w+):',l)
    if m: cur=[m.group(1),i,None]; funcs.append(cur)
    if cur and 's_endpgm' in l and cur[2] is None: cur[2]=i
def regs(tok):
    out=set()
    for m in re.finditer(r'\b([sva])\[(\d+):(\d+)\]',tok):
        for i in range(int(m.group(2)),int(m.group(3))+1): out.add(m.group(1)+str(i))
    for m in re.finditer(r'\b([sva])(\d+)\b',re.sub(r'[sva]\[\d+:\d+\]','',tok)):
        out.add(m.group(1)+m.group(2))
    return out
NODST=('s_cmp','s_cbranch','s_branch','s_waitcnt','s_barrier','s_nop','buffer_store','ds_write','global_store','s_setprio','s_sleep','s_endpgm','s_bitcmp','s_setreg','global_atomic','ds_add','ds_max','ds_min','flat_store','scratch_store','s_sendmsg','s_trap','s_icache','s_dcache','buffer_wbl2','buffer_inv','s_sethalt','ds_or','ds_and')
def analyse(name,a,b):
    insts=[]; labels={}; inloop=[]
    for i in range(a,b+1):
        l=text[i]; t=l.split(';')[0].rstrip()
        m=re.match(r'^(\.LBB\w+):',t)
        if m: labels[m.group(1)]=len(insts); continue
        t=t.strip()
        if not t or t.startswith('.') or t.endswith(':'): continue
        m=re.match(r'(\S+)\s*(.*)',t); op,args=m.group(1),m.group(2)
        parts=[x.strip() for x in args.split(',')] if args else []
        if op.startswith(NODST) or (op.startswith('buffer_load') and 'lds' in args.split()[-1:]):
            d=set(); u=set().union(*[regs(p) for p in parts]) if parts else set()
        elif op.startswith('v_mad_u64_u32') or op.startswith('v_mad_i64_i32'):
            d=regs(parts[0])|regs(parts[1]); u=set().union(*[regs(p) for p in parts[2:4]])|{sorted(regs(parts[4]))[0]} if len(parts)>4 and regs(parts[4]) else set().union(*[regs(p) for p in parts[2:4]])
        elif re.match(r'v_(add|sub|subrev)_co_u32_e64|v_(addc|subb|subbrev)_co_u32_e64|v_div_scale',op):
            d=regs(parts[0])|regs(parts[1]); u=set().union(*[regs(p) for p in parts[2:]])
        else:
            d=regs(parts[0]) if parts else set(); u=set().union(*[regs(p) for p in parts[1:]]) if len(parts)>1 else set()
            if op.startswith('v_writelane') or 'mfma' in op and False: u|=d
            if op.startswith('v_writelane'): u|=d
        tgt=parts[0] if op.startswith(('s_cbranch','s_branch')) else None
        insts.append((i+1,op,d,u,tgt)); inloop.append('in Loop' in l or False)
    n=len(insts)
    if n==0: return
    succ=[[] for _ in range(n)]
    for k,(ln,op,d,u,tgt) in enumerate(insts):
        if op=='s_endpgm': continue
        if op=='s_branch':
            if tgt in labels: succ[k].append(labels[tgt])
            continue
        if k+1<n: succ[k].append(k+1)
        if tgt and tgt in labels: succ[k].append(labels[tgt])
    pred=[[] for _ in range(n)]
    for k in range(n):
        for s_ in succ[k]: pred[s_].append(k)
    ALL=set()
    for ins in insts: ALL|=ins[2]|ins[3]
    entry={'s%d'%i for i in range(0,20)}|{'v0','v1','v2'}
    IN=[set(ALL) for _ in range(n)]; OUT=[set(ALL) for _ in range(n)]
    dq=collections.deque(range(n)); inq=[True]*n
    while dq:
        k=dq.popleft(); inq[k]=False
        newin=set(entry) if k==0 else (set.intersection(*[OUT[p] for p in pred[k]]) if pred[k] else set(ALL))
        newout=newin|insts[k][2]
        if newin!=IN[k] or newout!=OUT[k]:
            IN[k]=newin; OUT[k]=newout
            for s_ in succ[k]:
                if not inq[s_]: dq.append(s_); inq[s_]=True
    # loop membership: instruction index ranges between a label that is a back-edge target and the branch to it
    loops=[]
    for k,(ln,op,d,u,tgt) in enumerate(insts):
        if tgt in labels and labels[tgt]<=k: loops.append((labels[tgt],k))
    def in_loop(k): return any(a_<=k<=b_ for a_,b_ in loops)
    copies={}; others={}
    for k,(ln,op,d,u,tgt) in enumerate(insts):
        if not in_loop(k): continue
        for r in u:
            if r not in IN[k] and r[0] in 'va':
                (copies if op.startswith(('v_mov','v_accvgpr')) else others).setdefault(r,(ln,op))
    if copies or others:
        dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()[:110]
        print(f'{dn}\n    phi-copy reads of maybe-undefined registers in loops: {len(copies)} {sorted(copies)[:12]}\n    other reads: {len(others)} {sorted((r,v[1]) for r,v in others.items())[:8]}')
for name,a,b in funcs:
    if b: analyse(name,a,b)
