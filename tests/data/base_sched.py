# a config fragment written for the tests (not a reference file)
optimizer = dict(type='SGD', lr=0.02, momentum=0.9, weight_decay=0.0001)
lr_config = dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=0.001, step=[8, 11])
total_epochs = 12
model = dict(type='LSDetector', backbone=dict(type='ResNet', depth=50, norm_cfg=dict(type='BN', requires_grad=True)),
             neck=dict(type='FPN', num_outs=5))
