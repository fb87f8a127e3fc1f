"""Which torch.distributed settings let RCCL collectives be captured into a hipGraph on this stack?
One child process per setting (world size 1 on cuda:0 -- the box has one GPU), each with its own
timeout; a child that hangs is killed by handle.  Prints one line per setting.
usage: python tools/probes/rccl_graph_capture.py [--two-deep]
  --two-deep  also try the collectives issued two stream-forks away from the capturing stream (round 2:
              those never return from the capture on torch 2.10 / RCCL 2.26 / HIP 7.0 -- the child is killed
              by handle after 60 s without progress)"""
import multiprocessing as mp
import os
import socket
import sys

SETTINGS = {
    "defaults": {},
    "async_error_handling_off": {"TORCH_NCCL_ASYNC_ERROR_HANDLING": "0"},
    "watchdog_quiet": {"TORCH_NCCL_ASYNC_ERROR_HANDLING": "0", "TORCH_NCCL_ENABLE_MONITORING": "0",
                       "TORCH_NCCL_DUMP_ON_TIMEOUT": "0", "TORCH_NCCL_ENABLE_TIMING": "0"},
}


def child(name, env, port, q):
    os.environ.update(env)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    def say(m):
        q.put((name, m))
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        t = torch.ones(1 << 20, device=dev)
        a, o = torch.arange(4096, device=dev, dtype=torch.float32), torch.zeros(4096, device=dev)
        dist.all_reduce(t)                       # communicator up before any capture
        dist.all_to_all_single(o, a)
        torch.cuda.synchronize()
        say("eager ok")
        grp2 = dist.new_group()
        dist.all_to_all_single(o, a, group=grp2)
        torch.cuda.synchronize()
        fork = torch.cuda.Stream(dev)
        cases = ["all_reduce", "all_reduce_async", "all_to_all", "two_all_to_all", "all_to_all_other_group"]
        if "--two-deep" in sys.argv[1:]:
            cases += ["all_to_all_on_forked_stream", "all_to_all_other_group_on_forked_stream"]
        for what in cases:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    t.mul_(2.0)
                    if what == "all_reduce":
                        dist.all_reduce(t)
                    elif what == "all_reduce_async":
                        w = dist.all_reduce(t, async_op=True)
                        w.wait()
                    elif what == "all_to_all":
                        dist.all_to_all_single(o, a)
                    elif what == "two_all_to_all":
                        dist.all_to_all_single(o, a)
                        o.add_(1.0)
                        dist.all_to_all_single(a, o)
                    elif what == "all_to_all_other_group":
                        dist.all_to_all_single(o, a, group=grp2)
                    else:                            # a second-level fork: capture stream -> fork -> RCCL's stream
                        fork.wait_stream(side)
                        with torch.cuda.stream(fork):
                            o.mul_(1.0)
                            dist.all_to_all_single(o, a, group=grp2 if "other_group" in what else None)
                            o.add_(0.0)
                        t.add_(0.0)
                        side.wait_stream(fork)
                    t.add_(1.0)
            torch.cuda.current_stream(dev).wait_stream(side)
            say("captured " + what)
            before = float(t[0].item())
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            after = float(t[0].item())
            want = before
            for _ in range(3):
                want = want * 2.0 + 1.0
            say("replayed %s: %s" % (what, "values ok" if abs(after - want) < 1e-3 * abs(want) else "WRONG %g vs %g" % (after, want)))
            del g
        dist.destroy_process_group()
        say("done")
    except Exception as e:  # noqa: BLE001
        say("EXCEPTION " + repr(e)[:300])


def main():
    ctx = mp.get_context("spawn")
    for name, env in SETTINGS.items():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        q = ctx.Queue()
        p = ctx.Process(target=child, args=(name, env, port, q))
        p.start()
        last = "(nothing)"
        try:
            while True:
                n, m = q.get(timeout=60)
                last = m
                print("[%s] %s" % (n, m), flush=True)
                if m == "done" or m.startswith("EXCEPTION"):
                    break
        except Exception:  # noqa: BLE001 -- queue.Empty: no progress for 60 s
            print("[%s] HUNG after: %s" % (name, last), flush=True)
        p.join(timeout=15)
        if p.is_alive():
            p.kill()
            p.join(timeout=15)


if __name__ == "__main__":
    main()
