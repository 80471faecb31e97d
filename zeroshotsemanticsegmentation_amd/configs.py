"""configs.py -- the numbered experiment configurations of the reference (configs.py:1-143), same ids and values
(checked against tests/golden/configs.json, captured from the reference)."""


def _cfg(mode, dataset, embed_dim, fcn_epochs, train_unseen=(), val_unseen=(), fcn_lr=1e-5, fcn_loss='cos',
         fcn_optim='adam', seenmask_epochs=0, seenmask_lr=1e-3, **extra):
    d = dict(mode=mode, dataset=dataset, train_unseen=list(train_unseen), val_unseen=list(val_unseen),
             embed_dim=embed_dim, fcn_epochs=fcn_epochs, fcn_lr=fcn_lr, fcn_loss=fcn_loss, fcn_optim=fcn_optim,
             seenmask_epochs=seenmask_epochs, seenmask_lr=seenmask_lr)
    d.update(extra)
    return d


_PASCAL_8_2_10 = dict(train_unseen=[1, 13], val_unseen=[6, 7, 10, 14, 15, 16, 17, 18, 19, 20])
_PASCAL_16_2_2 = dict(train_unseen=[1, 13], val_unseen=[17, 19])
_CONTEXT_31_2_2 = dict(train_unseen=[0, 12], val_unseen=[16, 18])

configurations = {
    # FCN baseline with softmax inference (sum-reduced CE, hence the tiny learning rate)
    1: _cfg('train', 'pascal', 0, 30, fcn_lr=1e-10, fcn_loss='cross_entropy', fcn_optim='sgd'),
    # "one-hot" sized embedding space
    2: _cfg('train', 'pascal', 21, 30, one_hot_embed=False),
    # 20-d pascal
    4: _cfg('train', 'pascal', 20, 30),
    # 20-d 8/2/10 pascal zero-shot with seenmask: train / test
    14: _cfg('train', 'pascal', 20, 90, seenmask_epochs=10, **_PASCAL_8_2_10),
    15: _cfg('test_all', 'pascal', 20, 0, load_fcn_path="8_2_10_CFG_14_MODE_train_DATASET_pascal_TRAIN_UNSEEN_True_"
             "VAL_UNSEEN_True_EMBED_DIM_20_FCN_EPOCHS_90_FCN_LR_1e-05_FCN_LOSS_cos_FCN_OPTIM_adam_SEENMASK_EPOCHS_10_"
             "SEENMASK_LR_0.001_TIME_20180421-163751_", **_PASCAL_8_2_10),
    # 20-d 16/2/2 pascal zero-shot with seenmask: train / test
    16: _cfg('train', 'pascal', 20, 36, seenmask_epochs=10, **_PASCAL_16_2_2),
    17: _cfg('test_all', 'pascal', 20, 0, forced_unseen=False,
             load_fcn_path="16_2_2_CFG_16_MODE_train_DATASET_pascal_TRAIN_UNSEEN_True_VAL_UNSEEN_True_EMBED_DIM_20_"
             "FCN_EPOCHS_36_FCN_LR_1e-05_FCN_LOSS_cos_FCN_OPTIM_adam_SEENMASK_EPOCHS_10_SEENMASK_LR_0.001_"
             "TIME_20180421-163803_", **_PASCAL_16_2_2),
    # 20-d 31/2/2 context zero-shot with seenmask: train / test
    18: _cfg('train', 'context', 20, 59, seenmask_epochs=10, **_CONTEXT_31_2_2),
    19: _cfg('test_all', 'context', 20, 0, load_fcn_path="", **_CONTEXT_31_2_2),
}
