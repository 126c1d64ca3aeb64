import sys,re
for line in sys.stdin:
    if not line.startswith('W4TRACE'): 
        print(line.rstrip()[:200]); continue
    head, rest = line.split(':',1)
    ev = re.findall(r'(\d+)@(\d+)\(\+(\d+)\)', rest)
    print(head, ' '.join(f"{c}+{d}" for c,t,d in ev[:90]))
