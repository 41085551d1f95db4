import sys; sys.path.insert(0,'ransac-flow_amd')
import torch
from rfx.pipeline import cell_coords
bad=0
for (r,c) in [(30,40),(60,80),(50,66),(25,33),(15,20),(45,60),(50,165),(20,26),(33,44)]:
    Wd,Hd=cell_coords(r,c,torch.device('cuda:0')); Wc,Hc=cell_coords(r,c,torch.device('cpu'))
    nd=int((Wd.cpu()!=Wc).sum()+(Hd.cpu()!=Hc).sum()); bad+=nd
    print(r,c,"differing coords:",nd, float((Wd.cpu()-Wc).abs().max()), float((Hd.cpu()-Hc).abs().max()))
print("total differing", bad)
