# exercises _base_, nested merge, _delete_ and list replacement
_base_ = './base_sched.py'
optimizer = dict(lr=0.01)
lr_config = dict(step=[16, 22])
total_epochs = 24
model = dict(backbone=dict(_delete_=True, type='ResNeXt', depth=101, groups=64, base_width=4),
             neck=dict(norm_cfg=dict(type='GN', num_groups=32)))
