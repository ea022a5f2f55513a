python - <<'EOF'
import json
path='/root/.claude/projects/-root-repo/c1465933-34d3-4a14-8c00-199df833a45b.jsonl'
found=[]
for line in open(path):
    try: d=json.loads(line)
    except: continue
    msg=d.get('message',{})
    c=msg.get('content')
    if isinstance(c,list):
        for it in c:
            if it.get('type')=='tool_use' and it.get('name')=='Bash':
                cmd=it['input'].get('command','')
                if "new_kernel = r'''" in cmd or ("wsplit" in cmd and "PN_NET_GROUPS" in cmd):
                    found.append(cmd)
print(len(found))
for i,c in enumerate(found):
    open(f'/root/repo/tools/_tmp/patch{i}.sh','w').write(c)
    print(i,len(c))
EOF