import torch, time, json
dev=torch.device('cuda',0)
M=1024*197
x=torch.randn(M,768,device=dev).bfloat16(); w=(torch.randn(3072,768,device=dev)/28).bfloat16()
def win(n):
    st,en=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): torch.nn.functional.linear(x,w)
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en)/n*1e3
torch.nn.functional.linear(x,w); torch.cuda.synchronize()
time.sleep(2.0)
out=[round(win(20),1) for _ in range(40)]
print(json.dumps({"fc1_fwd_us_windows_of_20": out}))
time.sleep(3.0)
print(json.dumps({"after_3s_idle": [round(win(5),1) for _ in range(6)]}))
