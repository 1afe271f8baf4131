"""1F1B pipeline schedules as per-stage instruction streams.

`TrainSchedule.steps()` reproduces the instruction order of the reference's patched DeepSpeed schedule
(utils/patches.py:113-160: LoadMicroBatch first, then the P2P exchange, then compute) together with DeepSpeed's
step -> micro-batch index arithmetic; the order fixes the gradient-accumulation summation order, so it is kept
bit-exact (tests compare against oracle/intlogic.py and the worked S=2, M=4 example of SURVEY.md section 8(a2)).
`InferenceSchedule` is the forward-only stream used by eval_batch (train.py:183).
"""


class PipeInstruction:
    def __init__(self, **kwargs):
        self.name = self.__class__.__name__
        self.kwargs = kwargs
        for key, val in kwargs.items():
            setattr(self, key, val)

    def __repr__(self):
        args = ', '.join(f'{k}={v!r}' for k, v in self.kwargs.items())
        return f'{self.name}({args})'

    def __eq__(self, other):
        return type(self) is type(other) and self.kwargs == other.kwargs

    def __hash__(self):
        return hash((self.name, tuple(sorted(self.kwargs.items()))))


class OptimizerStep(PipeInstruction):
    pass


class ReduceGrads(PipeInstruction):
    pass


class ReduceTiedGrads(PipeInstruction):
    pass


class BufferOpInstruction(PipeInstruction):
    def __init__(self, buffer_id, **kwargs):
        super().__init__(buffer_id=buffer_id, **kwargs)


class LoadMicroBatch(BufferOpInstruction):
    pass


class ForwardPass(BufferOpInstruction):
    pass


class BackwardPass(BufferOpInstruction):
    pass


class SendActivation(BufferOpInstruction):
    pass


class RecvActivation(BufferOpInstruction):
    pass


class SendGrad(BufferOpInstruction):
    pass


class RecvGrad(BufferOpInstruction):
    pass


def _is_even(x):
    return x % 2 == 0


def _is_odd(x):
    return x % 2 != 0


class PipeSchedule:
    def __init__(self, micro_batches, stages, stage_id):
        self.micro_batches = micro_batches
        self.stages = stages
        self.stage_id = stage_id
        self.prev_stage = self.stage_id - 1
        self.next_stage = self.stage_id + 1

    def steps(self):
        raise NotImplementedError

    def num_pipe_buffers(self):
        return self.micro_batches

    def _valid_micro_batch(self, micro_batch_id):
        return 0 <= micro_batch_id < self.micro_batches

    def _valid_stage(self, stage_id):
        return 0 <= stage_id < self.stages

    @property
    def stage(self):
        return self.stage_id

    @property
    def num_stages(self):
        return self.stages

    @property
    def num_micro_batches(self):
        return self.micro_batches

    @property
    def is_first_stage(self):
        return self.stage_id == 0

    @property
    def is_last_stage(self):
        return self.stage_id == self.stages - 1

    def _buffer_idx(self, micro_batch_id):
        assert self._valid_micro_batch(micro_batch_id)
        return micro_batch_id % self.num_pipe_buffers()

    def __iter__(self):
        self.it = None
        return self

    def __next__(self):
        if self.it is None:
            self.it = self.steps()
        return next(self.it)


class InferenceSchedule(PipeSchedule):
    """Forward-only pipelining with two alternating buffers."""

    def steps(self):
        total_steps = self.micro_batches + self.stages - 1
        for step_id in range(total_steps):
            cmds = []
            micro_batch_id = step_id - self.stage_id
            if _is_even(self.stage_id):
                recv_buf, send_buf = step_id % 2, (step_id + 1) % 2
            else:
                recv_buf, send_buf = (step_id + 1) % 2, step_id % 2
            if self.is_first_stage or self.is_last_stage:
                if self._valid_micro_batch(micro_batch_id):
                    cmds.append(LoadMicroBatch(recv_buf))
            send = self._valid_stage(self.next_stage) and self._valid_micro_batch(micro_batch_id - 1)
            recv = self._valid_stage(self.prev_stage) and self._valid_micro_batch(micro_batch_id)
            if _is_even(self.stage_id):
                if send:
                    cmds.append(SendActivation(send_buf))
                if recv:
                    cmds.append(RecvActivation(recv_buf))
            else:
                if recv:
                    cmds.append(RecvActivation(recv_buf))
                if send:
                    cmds.append(SendActivation(send_buf))
            if self._valid_micro_batch(micro_batch_id):
                cmds.append(ForwardPass(recv_buf))
            yield cmds

    def num_pipe_buffers(self):
        return 2


class TrainSchedule(PipeSchedule):
    """1F1B: each stage alternates forward and backward micro-batches once the pipeline is full."""

    def steps(self):
        prev_micro_batch_id = -1
        total_steps = 2 * (self.micro_batches + self.stages - 1)
        for step_id in range(total_steps):
            micro_batch_id, is_forward = self._step_to_micro_batch(step_id)
            if self._valid_micro_batch(prev_micro_batch_id):
                prev_buffer = self._buffer_idx(prev_micro_batch_id)
            if self._valid_micro_batch(micro_batch_id):
                curr_buffer = self._buffer_idx(micro_batch_id)
            cmds = []
            # first / last stage pull their (features, label) micro-batch before any exchange
            if self.stage_id == 0 or self.stage_id == self.stages - 1:
                if is_forward and self._valid_micro_batch(micro_batch_id):
                    cmds.append(LoadMicroBatch(curr_buffer))
            if is_forward:
                if self._valid_micro_batch(prev_micro_batch_id) and self._valid_stage(self.prev_stage):
                    cmds.append(SendGrad(prev_buffer))
                if self._valid_micro_batch(micro_batch_id) and self._valid_stage(self.prev_stage):
                    cmds.append(RecvActivation(curr_buffer))
            else:
                if self._valid_micro_batch(micro_batch_id) and self._valid_stage(self.next_stage):
                    cmds.append(RecvGrad(curr_buffer))
                if self._valid_micro_batch(prev_micro_batch_id) and self._valid_stage(self.next_stage):
                    cmds.append(SendActivation(prev_buffer))
            if self._valid_micro_batch(micro_batch_id):
                cmds.append(ForwardPass(curr_buffer) if is_forward else BackwardPass(curr_buffer))
            if step_id == total_steps - 1:
                cmds.append(ReduceTiedGrads())
                cmds.append(ReduceGrads())
                cmds.append(OptimizerStep())
            prev_micro_batch_id = micro_batch_id
            yield cmds

    def num_pipe_buffers(self):
        return max(2, min(self.stages - self.stage_id, self.micro_batches))

    def _step_to_micro_batch(self, step_id):
        if _is_even(step_id) and _is_even(self.stage_id):
            return self._even_step_forward_id(step_id), True
        if _is_odd(step_id) and _is_odd(self.stage_id):
            return self._odd_step_forward_id(step_id), True
        if _is_even(step_id) and _is_odd(self.stage_id):
            return self._even_step_backward_id(step_id), False
        if _is_odd(step_id) and _is_even(self.stage_id):
            return self._odd_step_backward_id(step_id), False
        assert False

    def _even_step_forward_id(self, step_id):
        return int(step_id // 2 - self.stage_id // 2)

    def _odd_step_forward_id(self, step_id):
        return int((step_id - 1) // 2 - self.stage_id // 2)

    def _even_step_backward_id(self, step_id):
        return int(step_id // 2 - self.stages + (self.stage_id + 1) // 2)

    def _odd_step_backward_id(self, step_id):
        return int(((step_id - 1) // 2) - self.stages + 1 + self.stage_id // 2)


# ---- pipeline lanes: L independent TrainSchedules per stage, interleaved tick by tick (engine `pipe_lanes`)
def lane_micro_batches(micro_batches, lanes):
    """micro-batch i of a step runs on lane i % lanes: -> how many micro-batches each lane's schedule covers (lanes clipped to the micro-batch count)"""
    lanes = max(1, min(lanes, micro_batches))
    return [micro_batches // lanes + (1 if lane < micro_batches % lanes else 0) for lane in range(lanes)]


def interleave_lanes(schedules):
    """(lane, instructions) in issue order: tick t of every lane that still has one, lane 0 first, before tick t + 1 of any.

    TrainSchedule's step -> micro-batch arithmetic does not depend on the micro-batch count and every SendActivation / SendGrad it emits is received by the
    neighbour stage in the SAME tick, so two neighbours that both issue in this order post complementary sequences message by message: the property an
    in-order link needs (tests/test_intlogic.py simulates it under blocking receives and under full rendezvous)."""
    runs = [iter(s) for s in schedules]
    alive = list(range(len(runs)))
    while alive:
        for lane in list(alive):
            try:
                cmds = next(runs[lane])
            except StopIteration:
                alive.remove(lane)
                continue
            yield lane, cmds
