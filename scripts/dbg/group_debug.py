import sys, os
sys.path.insert(0,'ransac-flow_amd')
import torch
from rfx import weights, synth
from rfx.pipeline import AlignPipeline
dev=torch.device('cuda:0')
pipe=AlignPipeline(dict(trunk=weights.resnet50_trunk_sd(0)), nbScale=7, nbIter=1000, tolerance=0.05, minSize=640, scaleR=1.2, device=dev)
prep=pipe.prepare_device(*pipe.upload_raw([synth.make_pair(480,640,seed=0)]))
os.environ["RFX_GRAPH"]="0"
pipe.features(prep); torch.cuda.synchronize()
os.environ["RFX_GROUP_DEBUG"]="1"
