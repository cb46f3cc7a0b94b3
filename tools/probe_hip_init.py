import sys, ctypes as C, os
sys.path.insert(0, '/root/repo')
mode = sys.argv[1]
print("mode", mode, flush=True)
if mode != "notorch":
    import torch
    if mode == "avail":
        print("avail", torch.cuda.is_available())
    elif mode == "init":
        torch.cuda.init()
    elif mode == "count":
        print(torch.cuda.device_count())
    import oracle.clip_ref as R
    from knowledge_enhanced_multimodal_retrieval_amd import engine
    from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS
    eng = engine.ClipEngine(ARCHS["tiny"], "cuda:0")
    try:
        eng.load_state_dict(R.random_state_dict(R.ARCHS["tiny"]))
        print("finalize ok")
    except Exception as e:
        print("FAIL", e)
else:
    hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
    n = C.c_int(0)
    print("hipGetDeviceCount", hip.hipGetDeviceCount(C.byref(n)), n.value)
    p = C.c_void_p()
    print("hipMalloc", hip.hipMalloc(C.byref(p), 1024))
for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "HSA_ENABLE_IPC_MODE_LEGACY"):
    print(k, os.environ.get(k))
os.system("grep -c amdhip64 /proc/%d/maps; grep amdhip64 /proc/%d/maps | awk '{print $6}' | sort -u" % (os.getpid(), os.getpid()))
