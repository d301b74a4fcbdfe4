"""ORACLE (test infrastructure, NOT product code) -- the reference's epoch-loop bookkeeping restated literally from
/root/reference/train_utils/train_unet.py:312-323 and :460-523 (plotting removed), driven by given per-epoch losses.
Returns the lines the reference writes to its loss file (timing lines excluded) and the checkpoint names it saves.
The reference script itself cannot be imported (it parses argv, needs torch_ema and the dataset at import time)."""
import numpy as np


def run(train_losses, val_losses, test_losses, val_loss_SMA_window=10, validation_loss_count_threshold=5,
        train_indefinitely=False, save_at_epochs=(), weights_name="w", max_epochs=None):
    lines, saves = [], []
    validation_loss_increasing = False                      # :316
    prev_validation_loss = 0                                # :317
    e = 0                                                   # :319
    validation_loss_upward_counter = 0                      # :320
    validation_array = np.zeros(val_loss_SMA_window)        # :321
    min_validation_loss = 1000000                           # :323
    while not validation_loss_increasing:                   # :330
        train_loss, validation_loss, test_loss = train_losses[e], val_losses[e], test_losses[e]
        validation_array[e % val_loss_SMA_window] = validation_loss              # :460
        smoothed_validation_loss = np.mean(validation_array)                     # :462
        if smoothed_validation_loss > prev_validation_loss:                      # :465
            validation_loss_upward_counter += 1
        else:
            validation_loss_upward_counter = 0
        if validation_loss_upward_counter > validation_loss_count_threshold:     # :469
            validation_loss_increasing = True
            if train_indefinitely:
                lines.append(f'Validation loss stopped decreasing at epoch {e+1}')
                validation_loss_increasing = False
        prev_validation_loss = smoothed_validation_loss                          # :475
        if validation_loss < min_validation_loss:                                # :476
            lines.append('Validation loss is at a minimum. Saving the model')
            saves.append(weights_name + '.pth')
            min_validation_loss = validation_loss
        if train_indefinitely and len(save_at_epochs) > 0:                       # :484
            if e in save_at_epochs:
                saves.append(weights_name + '_epoch' + str(e) + '.pth')
        lines.append("[INFO] EPOCH: {}".format(e + 1))                           # :490
        lines.append("Train loss: {:.6f},  Validation loss: {:.6f}, Test loss: {:.6f}".format(
            train_loss, validation_loss, test_loss))
        e += 1                                                                   # :519
        if max_epochs is not None and e >= max_epochs:      # not in the reference: bounds train_indefinitely runs
            break
    lines.append('Training complete')                                            # :521
    return lines, saves, e
