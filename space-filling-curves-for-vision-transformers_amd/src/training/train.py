from sfcvit.training.loops import (cutmix_data, evaluate, mixup_criterion, mixup_data, rand_bbox, train,  # noqa: F401
                                   train_with_mixup_or_cutmix, train_with_scheduler)
