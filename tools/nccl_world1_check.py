import os, torch, torch.distributed as td
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.arange(65, dtype=torch.int64, device="cuda")
td.all_reduce(t); td.barrier()
objs = [None]; td.all_gather_object(objs, ("host", True)); lst = ["x"]; td.broadcast_object_list(lst, src=0)
print("nccl world-1 ok", int(t.sum()), objs, lst)
td.destroy_process_group()
