import sys; sys.path.insert(0,'.')
import numpy as np, swg_loader
swg=swg_loader.load()
sc=swg.load_scoring("PAM250"); ctx=swg.Context(0); ctx.set_scoring(sc,-3,-1); ctx.set_query(swg.synth_query(1,50))
flat,off=swg.synth_db(70+1023,1023,max_len=900)
db=swg.Database(flat,off).upload(ctx)
qs=[swg.synth_query(300+i,L) for i,L in enumerate((128,1,77,300,299,45,128))]
got,hits,st=ctx.search_multi(db,qs,k=4)
print({k:st[k] for k in ('cell_form','cols_per_wave','group_lanes','waves','long_pairs','long_cols_per_lane','workgroups')})
