import time, torch
torch.cuda.init()
g = torch.cuda.default_generators[0]
def t(fn, n=2000):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter()-t0)/n*1e6
print("get_offset", t(g.get_offset))
print("set_offset", t(lambda: g.set_offset(g.get_offset()+4)))
print("initial_seed", t(g.initial_seed))
print("default_generators[idx]", t(lambda: torch.cuda.default_generators[torch.cuda.current_device()]))
print("current_device", t(torch.cuda.current_device))
print("get_state", t(g.get_state, 200))
x=torch.zeros(4,device="cuda")
print("storage ptr", t(lambda: x.untyped_storage().data_ptr()))
torch.manual_seed(5); print(g.get_offset(), g.initial_seed())
g.set_offset(8); torch.manual_seed(5); print(g.get_offset())
