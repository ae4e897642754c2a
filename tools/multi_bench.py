import importlib, os, sys, json, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo") else os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5"); os.environ.setdefault("VSM_HOST_THREADS", "14")
import numpy as np, torch
import bench
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
print(json.dumps(bench.multi_sequence_leg(vm, synth, torch, torch.device("cuda:0"), 1242, 375), indent=1))
