"""Where the end-to-end pipeline loses against the GPU-only rate: run_shard with (a) a no-op consumer, (b) D2H only,
(c) D2H + log text, (d) + file write; and the prefetch thread on / off.  One GPU."""
import os, sys, time, tempfile, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import weights as W
from style_transfer_amd.attack import AttackConfig, AttackRunner
from style_transfer_amd.data import AttackSet, load_word_counts, load_word_dict_info
from style_transfer_amd.logfmt import LogWriter, VocabBytes, format_attack_block
from style_transfer_amd.pipeline import run_shard

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
dev = torch.device("cuda", 0)
ae = W.AEWeights(W.load_state_dict_file(os.path.join(GOLD, "surrogate_ae_yelp.safetensors")), dev, gemm_format="f16x3")
sc = W.ScorerWeights(W.load_state_dict_file(os.path.join(GOLD, "siamese_model_params.safetensors")),
                     W.load_state_dict_file(os.path.join(GOLD, "predictor_net_params.safetensors")), dev)
cfg = AttackConfig(not_nll=True, pos_victim_lr=0.2, neg_victim_lr=0.2, log_decodes=False, iters_per_decode=(12, 6))
runner = AttackRunner(ae, sc, cfg, dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 49152      # sentences per batch
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4         # batches
counts = load_word_counts(os.path.join(GOLD, "yelp", "word_to_id.txt"), ae.vocab)
data = AttackSet.synthetic(B * NB, ae.vocab, "yelp", seed=99, word_counts=counts)
with contextlib.redirect_stdout(sys.stderr):
    id_to_word, _ = load_word_dict_info(os.path.join(GOLD, "yelp", "word_to_id.txt"), 5)
vocab = VocabBytes(id_to_word)
td = tempfile.mkdtemp()


def run(name, consume, prefetch=True):
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        run_shard(runner, data, 0, len(data), B, consume, prefetch=prefetch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print("%-44s %8.0f sentences/s" % (name, len(data) / dt), flush=True)


# GPU only: batches prepared beforehand
preps = [runner.prepare(data.span(k * B, (k + 1) * B)) for k in range(NB)]
gidx = torch.arange(B, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    for p in preps:
        runner.run_prepared(p, gidx, fetch=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print("%-44s %8.0f sentences/s" % ("GPU only (inputs resident, no records)", B * NB / dt), flush=True)
del preps
run("pipeline, no-op consumer", lambda b0, b1, lean: None)
run("pipeline, D2H only", lambda b0, b1, lean: lean.to("cpu", pinned=True))
run("pipeline, D2H + log text", lambda b0, b1, lean: format_attack_block(vocab, data.span(b0, b1), lean.to("cpu", pinned=True), b0))
log = LogWriter(os.path.join(td, "log"), echo=False)
run("pipeline, D2H + log text + file write", lambda b0, b1, lean: log.write_block(format_attack_block(vocab, data.span(b0, b1), lean.to("cpu", pinned=True), b0)))
log.close()
run("no prefetch / drain threads, full consumer", lambda b0, b1, lean: format_attack_block(vocab, data.span(b0, b1), lean.to("cpu", pinned=True), b0), prefetch=False)
