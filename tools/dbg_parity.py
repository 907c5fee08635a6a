import numpy as np, demodulator_amd as dm, oracle, sys
orc=oracle.restatement()
for B,flags in ((1024,{}),(4096,{}),(262144,{}),(262144,dict(lowpass_in=12500.0,lowpass_out=6500.0))):
    opt=dm.Options(buf_size=B,**flags)
    data=oracle.synth_bytes(0xD3300D02,0,3*B if B>4096 else 8*B)
    want=orc.process(oracle.Flags(**opt.__dict__),data)
    with dm.Demodulator(opt) as d: got=d.process_host(data)
    bad=np.nonzero(np.abs(got-want)>1e-4*np.maximum(1,np.abs(want)))[0]
    print(B,flags,len(bad), bad[:24])
    for i in bad[:6]: print("   ",i, i%256, got[i], want[i])
