{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE RecordWildCards          #-}
-- |
-- Swim.Sim -- reference-side binding to libswimsim.so (include/swimsim.h).
--
-- SOURCE ONLY: this image has no GHC (SURVEY.md F4), so this module is delivered
-- uncompiled.  It imports the reference's own `Types` unchanged and offers the
-- tick simulator behind the same `Config` / `Gossip` / `Message` surface:
--
--   simulate :: SimConfig -> Source IO (Tick, Gossip)   -- events as `Broadcast (Suspect|Alive|Dead ..)`
--   stepN, memberView, firstDetection
--
-- Cabal stanza to add to swim.cabal's library:
--   exposed-modules: ..., Swim.Sim
--   extra-libraries: swimsim
--   include-dirs:    <repo>/include
--   build-depends:   ..., conduit, stm
module Swim.Sim
  ( SimConfig(..), Sim, Tick
  , defaultSimConfig, configureSim, stepN, scheduleFault
  , drainEvents, simulate, memberView, firstDetection, digest
  ) where

import           Control.Concurrent.MVar (MVar, newMVar, withMVar)
import           Control.Monad (forM, unless, when)
import           Control.Monad.IO.Class (liftIO)
import           Data.Conduit (Source, yield)
import           Data.Int (Int32, Int64)
import           Data.Word (Word16, Word32, Word64, Word8)
import           Foreign.C.String (CString, peekCString)
import           Foreign.C.Types (CInt (..), CSize (..))
import           Foreign.ForeignPtr (ForeignPtr, newForeignPtr, withForeignPtr)
import           Foreign.Marshal.Alloc (alloca, allocaBytes)
import           Foreign.Marshal.Array (allocaArray, peekArray)
import           Foreign.Ptr (FunPtr, Ptr, nullPtr, plusPtr)
import           Foreign.Storable (peek, peekByteOff, pokeByteOff)

import           Types  -- the reference's src/Types.hs, unchanged

type Tick = Word64

-- | Simulator knobs around the reference 'Config' (numToGossip and gossipInterval are
-- honoured; bindHost / joinHosts / udpBufferSize are carried unused, as in the reference).
data SimConfig = SimConfig
  { simCfg            :: Config
  , simMembers        :: Word32
  , simSeed           :: Word64
  , simLossPpm        :: Word32
  , simSuspicionTicks :: Word32   -- 0 = 3 * ceil(log2 N)
  , simRetransmitMult :: Word32   -- 0 = 3
  , simMaxSubjects    :: Word32
  , simDevice         :: Int32
  , simTargetScheme   :: Word32   -- 0 = kRandomMembers (the reference), 1 = robust round-robin (src/Core.hs:232 FIXME)
  }

data SwimsimT
newtype Sim = Sim (MVar (ForeignPtr SwimsimT))   -- one handle = one logical thread of control

-- struct swimsim_config: field offsets as laid out by include/swimsim.h (LP64)
cfgSize :: Int
cfgSize = 96

foreign import ccall unsafe "swimsim_default_config" c_default_config :: Ptr () -> IO CInt
foreign import ccall safe   "swimsim_create"         c_create  :: Ptr () -> Ptr (Ptr SwimsimT) -> IO CInt
foreign import ccall unsafe "&swimsim_destroy"       p_destroy :: FunPtr (Ptr SwimsimT -> IO ())
foreign import ccall unsafe "swimsim_last_error"     c_last_error :: Ptr SwimsimT -> IO CString
foreign import ccall safe   "swimsim_step"           c_step :: Ptr SwimsimT -> Word32 -> IO CInt   -- long-running => safe
foreign import ccall unsafe "swimsim_schedule_fault" c_fault :: Ptr SwimsimT -> Word64 -> Word32 -> Word8 -> IO CInt
foreign import ccall safe   "swimsim_drain_events"   c_drain :: Ptr SwimsimT -> Ptr () -> CSize -> Ptr CSize -> IO CInt
foreign import ccall safe   "swimsim_read_view"      c_view  :: Ptr SwimsimT -> Word32 -> Ptr () -> CSize -> Ptr CSize -> IO CInt
foreign import ccall safe   "swimsim_first_detect"   c_first :: Ptr SwimsimT -> Ptr Word64 -> CSize -> IO CInt
foreign import ccall safe   "swimsim_digest"         c_digest :: Ptr SwimsimT -> Ptr Word64 -> IO CInt
-- sharded clusters (one handle per GPU; include/swimsim.h "sharded clusters"): the caller moves the
-- records of the two exchange rounds between the handles (MPI / RCCL binding of the embedder's choice;
-- swim_amd/shard.py is the worked example over torch.distributed).
foreign import ccall unsafe "swimsim_shard_info"    c_shard_info    :: Ptr SwimsimT -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> IO CInt
foreign import ccall unsafe "swimsim_shard_buffers" c_shard_buffers :: Ptr SwimsimT -> Ptr (Ptr ()) -> Ptr (Ptr ()) -> IO CInt
foreign import ccall safe   "swimsim_shard_phase1"  c_shard_phase1  :: Ptr SwimsimT -> Ptr Word32 -> IO CInt
foreign import ccall safe   "swimsim_shard_phase2"  c_shard_phase2  :: Ptr SwimsimT -> Ptr Word32 -> Ptr Word32 -> IO CInt
foreign import ccall safe   "swimsim_shard_phase3"  c_shard_phase3  :: Ptr SwimsimT -> Ptr Word32 -> Ptr Word32 -> IO CInt

defaultSimConfig :: Config -> SimConfig
defaultSimConfig c = SimConfig c 128 1 0 0 0 0 0 0

memberNameOf :: Word32 -> String
memberNameOf i = 'm' : show i

-- | `configure :: IO (Either Error Store)` (src/Util.hs:103-107) for the whole population.
configureSim :: SimConfig -> IO (Either Error Sim)
configureSim SimConfig{..} =
  allocaBytes cfgSize $ \p -> alloca $ \ph -> do
    _ <- c_default_config p
    pokeByteOff p 8  (fromIntegral (numToGossip simCfg) :: Int32)
    pokeByteOff p 16 (fromIntegral (gossipInterval simCfg) :: Int64)
    pokeByteOff p 24 simMembers
    pokeByteOff p 32 simSeed
    pokeByteOff p 48 simLossPpm
    pokeByteOff p 52 simSuspicionTicks
    pokeByteOff p 56 simRetransmitMult
    pokeByteOff p 60 simMaxSubjects
    pokeByteOff p 80 simDevice
    pokeByteOff p 92 simTargetScheme
    rc <- c_create p ph
    if rc /= 0
      then Left <$> (c_last_error nullPtr >>= peekCString)
      else do h <- peek ph
              fp <- newForeignPtr p_destroy h
              Right . Sim <$> newMVar fp

withSim :: Sim -> (Ptr SwimsimT -> IO a) -> IO a
withSim (Sim mv) f = withMVar mv (`withForeignPtr` f)

check :: Ptr SwimsimT -> CInt -> IO ()
check h rc = unless (rc == 0) $ c_last_error h >>= peekCString >>= ioError . userError

-- | nticks periods of `failureDetector` for every member (src/Core.hs:236-240).
stepN :: Sim -> Word32 -> IO ()
stepN s n = withSim s $ \h -> c_step h n >>= check h

scheduleFault :: Sim -> Tick -> Word32 -> Bool -> IO ()
scheduleFault s t m up = withSim s $ \h -> c_fault h t m (if up then 1 else 0) >>= check h

-- swimsim_event_t: tick u64 @0, observer u32 @8, subject @12, incarnation @16, state u8 @20, cause u8 @21; 24 bytes
eventToGossip :: Ptr () -> IO (Tick, String, Gossip)
eventToGossip p = do
  t   <- peekByteOff p 0  :: IO Word64
  obs <- peekByteOff p 8  :: IO Word32
  sub <- peekByteOff p 12 :: IO Word32
  inc <- peekByteOff p 16 :: IO Word32
  st  <- peekByteOff p 20 :: IO Word8
  let name = memberNameOf sub
      msg = case st of
        1 -> Suspect { incarnation = fromIntegral inc, node = name }
        2 -> Dead { incarnation = fromIntegral inc, node = name, deadFrom = memberNameOf obs }
        _ -> Alive { incarnation = fromIntegral inc, node = name, addr = 0, port = 0 }
  return (t, memberNameOf obs, Broadcast msg)

-- | The `Broadcast` gossip the members enqueued since the last drain (src/Core.hs:119-121,254).
drainEvents :: Sim -> IO [(Tick, String, Gossip)]
drainEvents s = withSim s $ \h -> alloca $ \pn -> do
  rc0 <- c_drain h nullPtr 0 pn
  n <- peek pn
  if rc0 == 0 || n == 0 then return [] else
    allocaBytes (fromIntegral n * 24) $ \buf -> do
      c_drain h buf n pn >>= check h
      m <- peek pn
      forM [0 .. fromIntegral m - 1] $ \k -> eventToGossip (buf `plusPtr` (k * 24))

-- | Stream of membership events, one tick at a time, in the reference's `Source IO Gossip` style
-- (cf. `failureDetector :: Store -> Source IO Gossip`, src/Core.hs:233).
simulate :: Sim -> Word32 -> Source IO (Tick, Gossip)
simulate s ticks = go ticks
  where go 0 = return ()
        go k = do evs <- liftIO (stepN s 1 >> drainEvents s)
                  mapM_ (\(t, _, g) -> yield (t, g)) evs
                  go (k - 1)

-- | `members store` (src/Core.hs:76-77) for one observer: the non-default entries of its map.
-- swimsim_view_entry_t: subject @0, incarnation @4, since_tick @8, state u8 @12; 16 bytes
memberView :: Sim -> Word32 -> IO [(String, Liveness, Int, Tick)]
memberView s obs = withSim s $ \h -> alloca $ \pn -> do
  rc0 <- c_view h obs nullPtr 0 pn
  n <- peek pn
  if rc0 == 0 || n == 0 then return [] else
    allocaBytes (fromIntegral n * 16) $ \buf -> do
      c_view h obs buf n pn >>= check h
      forM [0 .. fromIntegral n - 1] $ \k -> do
        let p = buf `plusPtr` (k * 16)
        sub <- peekByteOff p 0 :: IO Word32
        inc <- peekByteOff p 4 :: IO Word32
        since <- peekByteOff p 8 :: IO Word32
        st <- peekByteOff p 12 :: IO Word8
        return (memberNameOf sub, toEnum (fromIntegral st), fromIntegral inc, fromIntegral since)

firstDetection :: Sim -> Word32 -> IO [Maybe Tick]
firstDetection s n = withSim s $ \h -> allocaArray (fromIntegral n) $ \buf -> do
  c_first h buf (fromIntegral n) >>= check h
  map (\v -> if v == maxBound then Nothing else Just v) <$> peekArray (fromIntegral n) buf

digest :: Sim -> IO Word64
digest s = withSim s $ \h -> alloca $ \p -> c_digest h p >>= check h >> peek p
