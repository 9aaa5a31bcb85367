import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden"))
import cases
from oracle import tps_oracle as O
O.build()
G = cases.load("warp_backward"); gi = cases.g14_inputs(); inp = cases.g2_inputs()
c = O.classic_constants(cases.CL_F, cases.CL_HW)
for chain in (False, True):
    o = O.warp_backward(gi["g_out_cl"], inp["img_smooth"], inp["ctrl"], c["inv_delta_C"], c["P_hat"], cases.CL_HW, chain_grid=chain)
    for k, g in (("g_in0", "cl_g_img"), ("g_ctrl", "cl_g_ctrl")):
        print("classic chain" if chain else "classic bmm  ", k, "max err", np.abs(o[k] - G[g]).max(), "scale", np.abs(G[g]).max())
inp = cases.g3_inputs(); c = O.tpspp_constants(cases.PP_HW, cases.PP_POINT)
kw = dict(P_xy=c["P_xy"], score=inp["score"], in1=inp["x"], g_out1=gi["g_out1"])
for chain in (False, True):
    o = O.warp_backward(gi["g_out0"], inp["feat_grid"], inp["ctrl"], c["hat_C"], c["P_hat"], cases.PP_HW, chain_grid=chain, **kw)
    for k, g in (("g_in0", "pp_g_feat_grid_sub"), ("g_in1", "pp_g_x_sub"), ("g_ctrl", "pp_g_ctrl"), ("g_score", "pp_g_score")):
        a = cases.sub(o[k]) if "in" in k else o[k]
        print("pp chain" if chain else "pp bmm  ", k, "max err", np.abs(a - G[g]).max(), "scale", np.abs(G[g]).max())
