"""Goldens of next-token selection, made by RUNNING THE REFERENCE's sample.py on the CPU
(``python tests/golden/gen_sampling_golden.py`` in the build container only).

For every case the reference's ``sample_next_ids`` is called row by row as its scheduler does
(scheduler.py:509-546); ``torch.multinomial`` is replaced for the call by a recorder, so the
golden holds the exact distribution the reference would draw from; greedy cases hold the ids.
"""
import os
import sys

import torch

sys.path.insert(0, "/root/reference")
import sample as ref  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    gen = torch.Generator().manual_seed(77)
    cases = []
    for vocab, rows in ((97, 5), (1000, 3)):
        logits = 3.0 * torch.randn((rows, vocab), generator=gen)
        logits[0, 5] = logits[0, 9] = logits[0].max() + 1.0  # a tie at the top
        prev = [torch.randint(0, vocab, (int(n),), generator=gen).tolist() for n in torch.randint(0, 12, (rows,), generator=gen)]
        for temperature, top_k, top_p, penalty in ((1.0, 0, 1.0, 1.0), (0.6, 0, 1.0, 1.0), (0.8, 10, 1.0, 1.0), (1.0, 0, 0.9, 1.0),
                                                   (0.7, 20, 0.85, 1.0), (1.3, 1, 0.5, 1.0), (0.0, 50, 0.3, 1.3), (0.9, 2000, 0.999, 1.1)):
            probs, greedy = [], []
            real = torch.multinomial
            try:
                for i in range(rows):
                    seen = {}
                    torch.multinomial = lambda p, num_samples, **kw: seen.setdefault("p", p.clone()) * 0 + torch.zeros(p.shape[:-1] + (1,), dtype=torch.long)
                    ref.sample_next_ids(logits[i : i + 1], True, temperature, top_k, top_p, prev[i], penalty)
                    probs.append(seen["p"])
                    greedy.append(ref.sample_next_ids(logits[i : i + 1], False, temperature, top_k, top_p, prev[i], penalty))
            finally:
                torch.multinomial = real
            cases.append({"logits": logits, "prev_ids": prev, "temperature": temperature, "top_k": top_k, "top_p": top_p,
                          "repetition_penalty": penalty, "probs": torch.cat(probs), "greedy_ids": torch.cat(greedy)})
    path = os.path.join(OUT, "sampling.pt")
    torch.save({"cases": cases}, path)
    print(f"sampling: {len(cases)} cases, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
