import sys; sys.path.insert(0,'/root/repo'); import pe_load
pe = pe_load.load()
eng = pe.ffi.Engine(); eng.set_options(g_min=0.0)
eng.load_deck(pe.deck.rc_mesh(100,100,1,True)); eng.reset()
eng.analyze_tr(1e-10, 3)
st = eng.analyze_tr(1e-10, 20)
print(st, eng.info()['n_fronts'])
