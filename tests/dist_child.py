"""Child process of tests/test_hip_dist.py: one rank of a torch.distributed "nccl" (= RCCL) job, started FRESH (nothing has touched the
GPU before init_process_group).  Start-up path of the multi-GPU deployment: model A is finalised from the state dict, model B never sees
it (finalize_empty) and receives the packed blob through ONE dist.broadcast; both run the same batch.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    from conformer_ocr_amd import synth
    from conformer_ocr_amd.dist import broadcast_weights, shard_batches
    from conformer_ocr_amd.engine import HipRecognizer
    hp = synth.hparams('cfg2', num_encoder_layers=2)
    image, lens, _, _ = synth.make_text_lines(4, hp.height, 400, seed=11)
    x = torch.from_numpy(image[:, 0]).to(dev)
    out = {'rank': rank, 'world': world}
    # 'padded': the reference's default width (144) -- in bf16 the blob that travels is the zero-padded 256-wide layout
    for dtype in ('bf16', 'fp32', 'padded'):
        if dtype == 'padded':
            hp, dtype_ = synth.hparams('cfg1', num_encoder_layers=2), 'bf16'
        else:
            dtype_ = dtype
        b = HipRecognizer(hp, dev, dtype_)
        b.finalize_empty()
        a = None
        if rank == 0:
            a = HipRecognizer(hp, dev, dtype_)
            a.load_state(synth.make_state_dict(hp, seed=3, decoder_gain=1.0, style='text'))
            a.finalize()
        broadcast_weights(b, src=0, source=a)
        lb, _ = b.forward(x, lens)
        torch.cuda.synchronize()
        h = float(lb.double().abs().sum().item())
        t = torch.tensor([h], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # every rank holds the same logits <=> max == own
        out[dtype] = {'blob_bytes': b.blob_nbytes(), 'finite': bool(torch.isfinite(lb).all()), 'same_on_all_ranks': float(t.item()) == h}
        if rank == 0:
            la, _ = a.forward(x, lens)
            torch.cuda.synchronize()
            out[dtype]['bit_equal_to_direct_finalize'] = bool(torch.equal(la, lb))
    # one data-parallel training step: the flat gradient vector is averaged over the ranks by one all-reduce
    from conformer_ocr_amd.codec import ascii_codec
    from conformer_ocr_amd.pred import PytorchRecognitionModel
    from conformer_ocr_amd.train import Trainer
    hpt = synth.hparams('tiny')
    st = synth.make_state_dict(hpt, seed=9, decoder_gain=1.0)
    net = PytorchRecognitionModel(**hpt.as_dict(), input_dropout_p=0.0, feed_forward_dropout_p=0.0, attention_dropout_p=0.0, conv_dropout_p=0.0,
                                  codec=ascii_codec(hpt.num_classes), compute_dtype='fp32')
    net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    net = net.to(dev).eval()
    im, ln = synth.make_lines(2, hpt.height, 64, seed=9 + rank, widths=[64, 40])
    batch = {'image': torch.from_numpy(im), 'seq_lens': torch.from_numpy(ln), 'target': torch.tensor([1, 2, 3]), 'target_lens': torch.tensor([2, 1])}
    tr = Trainer(net, lr=1e-3, distributed=True)
    l0 = tr.training_step(batch)
    l1 = tr.training_step(batch)
    out['train'] = {'loss0': l0, 'loss1': l1, 'grad_floats': int(tr.engine.train_grad_buffer().numel())}
    out['shard'] = shard_batches(5, rank, world)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
