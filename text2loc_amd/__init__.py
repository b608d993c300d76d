"""MI355X-native engine for Text2Loc's coarse text->cell retrieval path (and the rows around it: training step of the
3D-submap branch, PointNet++ object backbone, fine stage). The arithmetic lives in ``libt2l.so`` (hand-written HIP for
gfx950, C ABI in ``include/t2l.h``); these modules are the ctypes binding and the mirrors of the reference's Python
surface. Nothing here falls back to the CPU: without the library or without a GPU the mirrors raise ``T2LError``.

    engine          ctypes binding (Engine)                          cell_retrieval  CellRetrievalNetwork, LanguageEncoder
    coarse          eval_epoch, run_coarse, train_epoch              cross_matcher   CrossMatch, run_fine
    losses          ContrastiveLoss                                  optim           Adam
    packing         Object3d lists -> packed SoA, point batches      sharded         row- / query-sharded search over RCCL
    db              persistent CellDatabase                          synth           seeded synthetic weights / cells
"""

__all__ = ["cell_retrieval", "coarse", "cross_matcher", "db", "engine", "losses", "optim", "packing", "sharded", "synth"]
