import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/image-text-retrieval_amd")
import bench
from itr_amd import config as C, ops
from itr_amd.modalmodule import get_model
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg_file, ckpt, trans = bench.bert_files(os.path.join("/tmp", "itr_bench_bert"))
cfg = C.build_config(['with', 'CAMERA', 'data_name=coco_precomp'])
cfg.update(bert_config_file=cfg_file, init_checkpoint=ckpt, trans_cfg=trans, vocab_size=30522)
model = get_model(cfg); model.val_start()
feats, boxes, imgs_wh, ids, mask, types, lengths = bench.pooled_inputs(256, 1280, 'CAMERA', dev)
calls = []
orig_linear = ops.linear
def traced(x, w, b=None, act=None):
    y = orig_linear(x, w, b, act)
    calls.append((tuple(x.shape), tuple(w.shape), act, y))
    return y
res = {}
for mode in (False, True):
    ops.BF16X3 = mode
    calls.clear()
    ops.linear = traced
    import itr_amd.modalmodule.ImgEncoder as IE, itr_amd.modalmodule.camera_ as CM, itr_amd.modalmodule.TextEncoder as TE
    with torch.no_grad():
        img = model.img_enc(feats, boxes, imgs_wh)
        n_img_calls = len(calls)
        cap = model.txt_enc(ids[:256], mask[:256], types[:256])
    res[mode] = (img[0].clone(), cap.clone(), [(c[0], c[1], c[2], c[3].clone()) for c in calls], n_img_calls)
ops.linear = orig_linear
a, b = res[False], res[True]
print("img_emb max diff %.2e  cap_emb max diff %.2e" % ((a[0]-b[0]).abs().max().item(), (a[1]-b[1]).abs().max().item()))
for i, (ca, cb) in enumerate(zip(a[2], b[2])):
    d = (ca[3]-cb[3]).abs().max().item(); m = ca[3].abs().max().item()
    if i < a[3] or i > len(a[2]) - 12:
        print("%3d %s x %s act=%s  |y|max %.2e  diff %.2e  rel %.1e" % (i, ca[0], ca[1], ca[2], m, d, d / max(m, 1e-30)))
