"""CLI with the flags, defaults, file layout and log format of the reference's method/yelp/main.py
(/root/reference/method/yelp/main.py:35-126, __main__ :427-493) for the `--attack` path:

    cd method/yelp && python -m style_transfer_amd.main --attack --not_nll --pos_victim_lr=0.2 --neg_victim_lr=0.2

reads  ../../data/<task>/processed_files/{word_to_id.txt,neg10_pos10_test{0,1}.json}
       save/origin/{constraint_ae_model0_params.pkl,siamese_model_params.pkl,predictor_net_params.pkl}
writes ./loggings/<victim_optimizer>+<victim_lr>+iter<attack_max_iter>[date time]   (the *_log file
       outputs/get_i_iter.py parses).

Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N -m style_transfer_amd.main ...`;
the sentence list is split into N contiguous shards (one process per GPU, no communication while
computing), the per-sentence records are gathered to rank 0 (RCCL all_gather over xGMI) and rank 0
writes the log in global sentence order.  Training (`main.py` without --attack) is out of scope.
"""
import argparse
import os
import sys
import time

import numpy as np

MY_VERSION = 1.242  # main.py:25, echoed in the banner


def build_parser():
    p = argparse.ArgumentParser(description="Here is your model discription.")
    # main.py:36-39
    p.add_argument('--id_pad', type=int, default=0)
    p.add_argument('--id_unk', type=int, default=1)
    p.add_argument('--id_bos', type=int, default=2)
    p.add_argument('--id_eos', type=int, default=3)
    # main.py:44-47
    p.add_argument('--task', type=str, default='yelp')
    p.add_argument('--word_to_id_file', type=str, default='')
    p.add_argument('--data_path', type=str, default='')
    # main.py:52-60 (VAE parameters: parsed, unused by --attack)
    p.add_argument('--encoder_emb_dim', type=int, default=256)
    p.add_argument('--decoder_emb_dim', type=int, default=256)
    p.add_argument('--encoder_hidden_dim', type=int, default=256)
    p.add_argument('--decoder_hidden_dim', type=int, default=256)
    p.add_argument('--encoder_dropout', type=float, default=0.5)
    p.add_argument('--decoder_dropout', type=float, default=0.5)
    p.add_argument('--AE_LATENT_NUM', type=int, default=80)
    p.add_argument('--max_epoch', type=int, default=80)
    # main.py:66-77
    p.add_argument('--word_dict_max_num', type=int, default=5)
    p.add_argument('--batch_size', type=int, default=128)
    p.add_argument('--max_sequence_length', type=int, default=60)
    p.add_argument('--num_layers_AE', type=int, default=2)
    p.add_argument('--transformer_model_size', type=int, default=256)
    p.add_argument('--transformer_ff_size', type=int, default=1024)
    p.add_argument('--latent_size', type=int, default=256)
    p.add_argument('--word_dropout', type=float, default=1.0)
    p.add_argument('--embedding_dropout', type=float, default=0.5)
    p.add_argument('--learning_rate', type=float, default=0.001)
    p.add_argument('--label_size', type=int, default=1)
    # main.py:82-94
    p.add_argument('--train_siamese_lr', type=float, default=3e-3)
    p.add_argument('--simplest_siamese', action='store_true')
    p.add_argument('--negative_samples', action='store_true')
    p.add_argument('--tfidf_train', action='store_true')
    p.add_argument('--anneal_epoch', type=int, default=100)
    p.add_argument('--use_min', action='store_true')
    p.add_argument('--tau', type=float, default=0.1)
    p.add_argument('--bata', type=float, default=0.1)
    p.add_argument('--together', action='store_true')
    # main.py:100-124
    p.add_argument('--victim_optimizer', type=str, default='Adam')
    p.add_argument('--victim_lr', type=float, default=0.25)
    p.add_argument('--attack_max_iter', type=int, default=18)
    p.add_argument('--attack_max_change', type=int, default=7)
    p.add_argument('--reverse_order', action='store_true')
    p.add_argument('--diff', action='store_true')
    p.add_argument('--just_use_one', action='store_true')
    p.add_argument('--attack', action='store_true')
    p.add_argument('--retext', action='store_true')
    p.add_argument('--not_nll', action='store_true')
    p.add_argument('--early_stop', action='store_true')
    p.add_argument('--dance', action='store_true')
    p.add_argument('--pos_num', type=int, default=None)
    p.add_argument('--neg_num', type=int, default=None)
    p.add_argument('--pos_victim_lr', type=float, default=0.25)
    p.add_argument('--neg_victim_lr', type=float, default=0.25)
    # build-side options (no reference counterpart; defaults reproduce the reference's behaviour)
    g = p.add_argument_group("MI355X build options")
    g.add_argument('--attack_batch', type=int, default=256, help='sentences processed together per GPU')
    g.add_argument('--save_path', type=str, default=None, help="checkpoint directory (default save/origin/)")
    g.add_argument('--ae_checkpoint', type=str, default='constraint_ae_model0_params.pkl')
    g.add_argument('--limit', type=int, default=None, help='only the first N sentences of each test file')
    g.add_argument('--log_file', type=str, default=None, help='log path (default ./loggings/<reference name>)')
    g.add_argument('--no_log_decodes', action='store_true', help='skip the Rec/Sim/Ref decodes (lines omitted)')
    g.add_argument('--fixed_decode_steps', action='store_true', help='always run max_sequence_length-1 decode steps')
    g.add_argument('--gemm_format', default='f16x3', choices=['f16x3', 'f32'],
                   help='autoencoder matrix products: f16x3 = f16 hi+lo split operands, 3 MFMA products per term, f32 '
                        'accumulate (f32-grade results, default); f32 = exact f32 MFMA (~2x slower end to end)')
    g.add_argument('--quiet', action='store_true', help='do not echo log records on stdout')
    g.add_argument('--siamese_epochs', type=int, default=50, help='epochs of scorer training without --attack (main.py:490)')
    g.add_argument('--train_files', type=str, default='neg10_pos10_train0.json,neg10_pos10_train1.json',
                   help='training records under --data_path (siamese_contrastive.py:234-239; same schema as the test files)')
    g.add_argument('--init_seed', type=int, default=None, help='seed of the fresh SiameseNet / Predictor weights')
    g.add_argument('--train_ae', action='store_true',
                   help="run train_ae_model_constraint (main.py:470-471: the branch taken when no checkpoint is loaded) on "
                        "ae_train_constraint{0,1}.json under --data_path, starting from --ae_checkpoint when it exists, else "
                        "from make_model's initialisation; writes constraint_ae_model<epoch>_params.pkl under --save_path")
    g.add_argument('--ae_epochs', type=int, default=1, help='epochs of --train_ae (train_ae_model_constraint default: 1)')
    return p


def dist_info():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


from .pipeline import gather_to_rank0, shard_range  # noqa: E402,F401  (shard_range is part of this module's API)


def gather_dense_to_rank0(res, rank, world, device):
    """--dance / trace runs keep the dense records (two iterations, small runs): every field incl. the dance loss and
    scores and the final latents goes to rank 0 with the same exact-size send / recv as the lean records."""
    import torch
    import torch.distributed as dist
    from .attack import AttackResult
    fields = ["n_iter", "init_ids", "iter_ids", "changed", "like_ids", "unlike_ids", "final_latent"]
    vals = {f: getattr(res, f) for f in fields}
    if res.dance:
        vals["dance_loss"], vals["dance_scores"] = res.dance["loss"], res.dance["scores"]
    names = fields + ["dance_loss", "dance_scores"]
    ten = {f: None if vals.get(f) is None else torch.from_numpy(np.ascontiguousarray(vals[f])).to(device) for f in names}
    mine = torch.tensor([-1 if ten[f] is None else ten[f].shape[0] for f in names], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes = torch.stack(sizes).cpu().tolist()
    present = [all(sz[i] >= 0 for sz in sizes) for i in range(len(names))]
    if rank != 0:
        for i, f in enumerate(names):
            if present[i] and ten[f].shape[0] > 0:
                dist.send(ten[f], 0)
        return None
    out = {}
    for i, f in enumerate(names):
        if not present[i]:
            out[f] = None
            continue
        parts = [ten[f]]
        for r in range(1, world):
            buf = torch.empty((sizes[r][i],) + tuple(ten[f].shape[1:]), dtype=ten[f].dtype, device=device)
            if sizes[r][i] > 0:
                dist.recv(buf, r)
            parts.append(buf)
        out[f] = torch.cat(parts, 0).cpu().numpy()
    full = AttackResult(*[out[f] for f in fields])
    if res.dance:
        full.dance = dict(res.dance, loss=out["dance_loss"], scores=out["dance_scores"])
    return full


def train_scorer_main(args, ae, vocab_size, data_path, save_path, device, rank, world, log):
    """`python main.py` without --attack (main.py:488-490): train SiameseNet + Predictor on the frozen AE's latents
    and write siamese_model_params.pkl / predictor_net_params.pkl under save_path after every epoch."""
    import torch
    from .attack import AttackConfig, AttackRunner
    from .data import AttackSet
    from .train_scorer import ScorerTrainer, TrainConfig, fresh_state_dicts, train_contrastive
    files = [os.path.join(data_path, f) for f in args.train_files.split(',')]
    say = log if rank == 0 else (lambda *_: None)
    say('Loading train Dataset, Shuffle=True')
    say('ori_file_list = {}'.format(files))
    if rank == 0:
        print('Random Seed is set to 234')
    data = AttackSet.from_json(files, vocab_size, args.id_unk, args.max_sequence_length, args.pos_num, args.neg_num)
    if data.n_pos < 1 or data.n_neg < 1:
        raise ValueError("training needs at least one like and one unlike sentence per record (NLL form, siamese_contrastive.py:438-446)")
    runner = AttackRunner(ae, None, AttackConfig.from_namespace(args), device)
    ori, like, unlike = [], [], []
    CH = 8192                                    # records per encoder pass (21 sentences each)
    for c0 in range(0, len(data), CH):
        sub = data.subset(list(range(c0, min(len(data), c0 + CH))))
        lat = runner.encode_prepared(runner.prepare(sub))
        n = len(sub)
        ori.append(lat[:n].clone())
        like.append(lat[n:n + n * data.n_pos].view(n, data.n_pos, 256).clone())
        unlike.append(lat[n + n * data.n_pos:].view(n, data.n_neg, 256).clone())
    ori, like, unlike = torch.cat(ori), torch.cat(like), torch.cat(unlike)
    s_sd, p_sd = fresh_state_dicts(args.init_seed)
    if world > 1:                                # every rank starts from rank 0's weights
        import torch.distributed as dist
        for sd in (s_sd, p_sd):
            for k in sorted(sd):
                t = sd[k].to(device)
                dist.broadcast(t, 0)
                sd[k] = t.cpu()
    trainer = ScorerTrainer(s_sd, p_sd, TrainConfig(lr=args.train_siamese_lr, tau=args.tau, simplest=args.simplest_siamese,
                                                    just_use_one=args.just_use_one), device)
    os.makedirs(save_path, exist_ok=True)
    train_contrastive(trainer, ori, like, unlike, batch_size=args.batch_size, epochs=args.siamese_epochs, log=say,
                      rank=rank, world=world, save_path=save_path)
    return 0


def train_ae_main(args, vocab_size, data_path, save_path, device, rank, world, log):
    """--train_ae: the autoencoder's constraint training (siamese_contrastive.py:1000-1088) on the HIP step"""
    from . import weights as W
    from .data import AttackSet
    from .train_ae import AETrainConfig, AETrainer, fresh_ae_state_dict, train_ae_model_constraint
    if world > 1:
        raise SystemExit("--train_ae runs on one GPU (the in-batch constraint couples the rows of a batch; the N-GPU form of "
                         "the similarity block is contrastive.constraint_loss_and_grads)")
    ck = os.path.join(save_path, args.ae_checkpoint)
    if os.path.exists(ck):
        sd = W.expand_ae_state_dict(W.load_state_dict_file(ck))
    else:
        sd = fresh_ae_state_dict(vocab_size, args.num_layers_AE, args.transformer_model_size, args.transformer_ff_size, args.init_seed)
    files = [os.path.join(data_path, 'ae_train_constraint0.json'), os.path.join(data_path, 'ae_train_constraint1.json')]
    log('Loading Dataset, Shuffle=True')
    log('ori_file_list = {}'.format(files))
    print('Random Seed is set to 234')
    if args.batch_size % 32 != 0 or args.batch_size <= 0:
        raise SystemExit("--train_ae needs --batch_size to be a positive multiple of 32 (the in-batch constraint kernel); got %d" % args.batch_size)
    # create_ae_data_loader calls create_batches with pos_num=None / neg_num=None (siamese_contrastive.py:950-995): the number
    # of like views always comes from the file's sim_num, whatever --pos_num says
    data = AttackSet.from_json(files, vocab_size, args.id_unk, args.max_sequence_length, None, 0)
    trainer = AETrainer(sd, AETrainConfig(bata=args.bata, tau=args.tau), device)
    os.makedirs(save_path, exist_ok=True)
    train_ae_model_constraint(trainer, data, args.batch_size, args.ae_epochs, log, save_path)
    log.close()
    print('Done')
    return 0


def main(argv=None):
    args = build_parser().parse_args(argv)
    args.if_load_from_checkpoint = True        # main.py:130-131
    args.checkpoint_name = "origin"
    import torch
    from . import weights as W
    from .attack import AttackConfig, AttackRunner, concat_results, empty_result, write_attack_log
    from .pipeline import run_shard
    from .data import AttackSet, load_word_dict_info
    from .logfmt import LogWriter, headline

    rank, world, local = dist_info()
    if not torch.cuda.is_available():
        sys.exit("no GPU visible: the attack path runs on HIP kernels only (no CPU fallback)")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    # initialize_logging (siamese_contrastive.py:1595-1605): the name uses --victim_lr, not pos/neg lr
    log = None
    if rank == 0:
        now_time = time.strftime("[%Y-%m-%d %H%M%S]", time.localtime())
        path = args.log_file
        if path is None:
            os.makedirs('./loggings', exist_ok=True)
            path = os.path.join('./loggings', args.victim_optimizer + '+' + str(args.victim_lr) + '+iter' + str(args.attack_max_iter) + now_time)
        print('Saving Siamese Log to {}'.format(path))
        os.makedirs('./outputs', exist_ok=True)  # initialize_output, :1608-1611
        print('--- Running Siamese Style Transfer, Version{}'.format(MY_VERSION))
        print('--- Cuda: {}'.format(True))
        log = LogWriter(path, echo=not args.quiet)
        # print_basic_information (main.py:413-425)
        log('Path: {}'.format(os.path.abspath('.')))
        log('--- Checkpoint Path: {}'.format(args.checkpoint_name))
        log('--- AE Model Name: {}'.format(args.ae_checkpoint))
        log('--- Attack: {}'.format(args.attack))
        log('--- pos_num: {}'.format(args.pos_num))
        log('--- neg_num: {}'.format(args.neg_num))
        log('--- pos_victim_lr: {}'.format(args.pos_victim_lr))
        log('--- neg_victim_lr: {}'.format(args.neg_victim_lr))
        log('--- not_nll: {}'.format(args.not_nll))
        log('--- simplest_siamese: {}'.format(args.simplest_siamese))
        log(str(args))

    # preparation (main.py:153-199)
    save_path = args.save_path or 'save/%s/' % args.checkpoint_name
    if args.task not in ('yelp', 'amazon'):
        raise TypeError('Wrong task type!')
    data_path = args.data_path or '../../data/%s/processed_files/' % args.task
    if rank == 0:
        print("prepare data ...")
    id_to_word, vocab_size = load_word_dict_info(os.path.join(data_path, 'word_to_id.txt'), args.word_dict_max_num)

    if args.train_ae:
        return train_ae_main(args, vocab_size, data_path, save_path, device, rank, world, log)
    ae_sd = W.load_state_dict_file(os.path.join(save_path, args.ae_checkpoint))
    ae = W.AEWeights(ae_sd, device, max_len=args.max_sequence_length, gemm_format=args.gemm_format)
    if ae.vocab != vocab_size:
        raise ValueError("checkpoint vocabulary %d != word_to_id vocabulary %d" % (ae.vocab, vocab_size))
    if not args.attack:
        # main.py:488-490: train_contrastive(siamese_net, predictor_net, ae_model, args, 50) on fresh nets
        rc = train_scorer_main(args, ae, vocab_size, data_path, save_path, device, rank, world, log)
        if rank == 0:
            log.close()
            print('Done')
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return rc
    # load_siamese (siamese_contrastive.py:1565-1587): exit(0) when a file is missing
    sp, pp = os.path.join(save_path, 'siamese_model_params.pkl'), os.path.join(save_path, 'predictor_net_params.pkl')
    # the same state-dicts re-saved as safetensors (tests/golden) are accepted when the .pkl is absent
    alt = lambda q: q if os.path.exists(q) or not os.path.exists(q[:-4] + '.safetensors') else q[:-4] + '.safetensors'
    sp, pp = alt(sp), alt(pp)
    if rank == 0:
        print('Loading Siamese Net from {}'.format(save_path))
    if not os.path.exists(sp):
        print('Siamese Net Path not Exist')
        sys.exit(0)
    if not os.path.exists(pp):
        print('Predictor Net Path Not Exist')
        sys.exit(0)
    scorer = W.ScorerWeights(W.load_state_dict_file(sp), W.load_state_dict_file(pp), device, gemm_format=args.gemm_format)

    files = [os.path.join(data_path, 'neg10_pos10_test0.json'), os.path.join(data_path, 'neg10_pos10_test1.json')]
    if args.reverse_order:
        files = files[::-1]
    if rank == 0:
        log('Loading attack Dataset, Shuffle=False')
        log('ori_file_list = {}'.format(files))
        print('Random Seed is set to 234')
    data = AttackSet.from_json(files, vocab_size, args.id_unk, args.max_sequence_length, args.pos_num, args.neg_num)
    if args.limit is not None:
        half = len(data) // 2
        keep = list(range(min(args.limit, half))) + list(range(half, half + min(args.limit, half)))
        data = data.subset(keep)
    if rank == 0:
        log(headline('Start Attack via Contrastive Net'))
        log(headline('Optimizer {}, lr {}, Iter {}'.format(args.victim_optimizer, args.victim_lr, args.attack_max_iter)))
        log(headline('Attack with Simplest Siamest Net' if args.simplest_siamese else 'Attack with enhance Siamese Net(with Predictor)', note_token='+'))

    cfg = AttackConfig.from_namespace(args, log_decodes=not args.no_log_decodes, early_exit=not args.fixed_decode_steps)
    runner = AttackRunner(ae, scorer, cfg, device)
    runner.set_vocab(id_to_word)
    lo, hi = shard_range(len(data), rank, world)
    t0 = time.perf_counter()
    if args.dance:
        # --dance prints per-iteration losses and score digests (siamese_contrastive.py:833-835): dense records, the
        # Python writer (two iterations per sentence: small by construction)
        parts = []
        for b0 in range(lo, hi, args.attack_batch):
            b1 = min(hi, b0 + args.attack_batch)
            parts.append(runner.run_batch(data.span(b0, b1), list(range(b0, b1))))
        res = concat_results(parts) if parts else empty_result(cfg, data.n_pos, data.n_neg)   # more ranks than sentences
        if not parts:
            res.dance = dict(loss=np.zeros((0, 2), np.float32), scores=np.zeros((0, 2, data.n_pos + data.n_neg), np.float32),
                             n_pos=data.n_pos, not_nll=cfg.not_nll)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        if world > 1:
            res = gather_dense_to_rank0(res, rank, world, device)
        if rank == 0:
            write_attack_log(log, data, res, id_to_word, 0, dance=True)
    else:
        from .attack import LeanRecord
        from .logfmt import VocabBytes, format_attack_block
        vocab = VocabBytes(id_to_word) if rank == 0 else None
        fmt = lambda b0, b1, rec: format_attack_block(vocab, data.span(b0, b1), rec, b0, args.id_eos, args.id_pad, args.id_unk)
        kept = []
        if world == 1:
            # one GPU: records are downloaded, formatted and written batch by batch while the next batch computes
            consume = lambda b0, b1, lean: log.write_block(fmt(b0, b1, lean.to("cpu", pinned=True)))
        else:
            consume = lambda b0, b1, lean: kept.append(lean)
        run_shard(runner, data, lo, hi, args.attack_batch, consume)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        if world > 1:
            mine = LeanRecord.cat(kept) if kept else LeanRecord.empty(cfg, data.n_pos, data.n_neg, device)
            full = gather_to_rank0(mine, rank, world, device)
            if rank == 0:
                from .logfmt import format_attack_blocks
                for block in format_attack_blocks(vocab, data, full.cpu(), 0, eos=args.id_eos, pad=args.id_pad, unk=args.id_unk):
                    log.write_block(block)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if rank == 0:
        log.close()
        print('Done')
        print('[st] %d sentences on %d GPU(s): %.2f s compute on rank 0 (%.1f sentences/s)' %
              (len(data), world, dt, (hi - lo) / max(dt, 1e-9) * world), file=sys.stderr)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
