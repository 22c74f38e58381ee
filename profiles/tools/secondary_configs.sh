# End-of-round numbers of the secondary configurations (one line each): bash profiles/tools/secondary_configs.sh
B="python bench.py --steps 30 --warmup 5 --cpu-baseline-steps 0 --no-probe"
run() { name="$1"; shift; $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('| \`$name\` |', round(d['value']), '|', round(d['ms_per_step'],2), '|', round(d['step_roofline_frac'],2), '|', flush=True)"; }
run "(headline: 3B + 1B, K=4, 1 row, bf16)"
run "--weight-dtype fp8" --weight-dtype fp8
run "--target llama-3-8b" --target llama-3-8b
run "--target llama-3-8b --weight-dtype fp8" --target llama-3-8b --weight-dtype fp8
run "--batch 8" --batch 8
run "--batch 4 --target llama-3-8b (config 4 per GPU)" --batch 4 --target llama-3-8b
run "--draft-mode medusa-heads" --draft-mode medusa-heads
run "--draft-mode medusa-heads --target llama-3-8b --weight-dtype fp8" --draft-mode medusa-heads --target llama-3-8b --weight-dtype fp8
run "--draft-mode medusa (tied heads)" --draft-mode medusa
run "--draft-mode eagle" --draft-mode eagle
run "--do-sample (sampled bonus token)" --do-sample
