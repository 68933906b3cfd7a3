import subprocess, os, sys, tempfile
sys.path.insert(0, ".")
from kmer_id_amd import _build, synth
tool=_build.cli_path("kid_synth_files"); gz=_build.cli_path("kid_gzcat")
d=tempfile.mkdtemp()
parent,cnt=synth.load_taxonomy("bact10")
open(d+"/c.txt","w").write("".join("%d,%d\n"%(t,c) for t,c in enumerate(cnt.tolist())))
open(d+"/t.txt","w").write("".join("%d\t%d\n"%(x,y) for y,x in enumerate(parent.tolist()) if y>=2 and x!=1))
for lvl in ("1","6"):
    os.makedirs(d+"/f"+lvl)
    subprocess.check_call([tool,"fastq","--counts",d+"/c.txt","--tree",d+"/t.txt","--out-dir",d+"/f"+lvl+"/","--samples","1","--pairs","1000000","--level",lvl])
    subprocess.check_call([tool,"probes","--counts",d+"/c.txt","--out",d+"/p"+lvl+".gz","--scale","0.2","--level",lvl])
    for f in (d+"/f"+lvl+"/S0_R1_tr.fastq.gz", d+"/p"+lvl+".gz"):
        print(lvl, subprocess.run([gz,"--time","--zlib",f],stdout=subprocess.PIPE).stdout.decode().strip(), flush=True)
        print(lvl, subprocess.run([gz,"--time","--room","8388608",f],stdout=subprocess.PIPE).stdout.decode().strip(), flush=True)
        for t in (2,4,8,16):
            for ch in (1<<20, 2<<20, 4<<20):
                print(lvl, "threads", t, "chunk", ch, subprocess.run([gz,"--time","--threads",str(t),"--chunk",str(ch),"--room","8388608",f],stdout=subprocess.PIPE).stdout.decode().strip(), flush=True)
